#!/bin/bash
# GPU box, round 5, pass O: the parked stepper of n = 13..16 with the trial state / error combination waiting in the LDS rows of y /
# dydt across the last right-hand side (HAMK_RKF_SWAP_LAST=1, the new default) against the round-4 placement (=0), stepHam lines,
# same box back to back; then the stepper's GPU tests.
set -u
export HAMK_TEST_OVERRIDES=1
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export HAMK_CACHE_DIR=$PWD/.hamk_cache
O=gpurun_out; mkdir -p $O
: > $O/r05o_swap_last_ab.jsonl
for rep in 1 2; do
for s in chain13 chain14 chain16; do
  for f in "-DHAMK_RKF_SWAP_LAST=0" ""; do
    HAMK_HIPRTC_FLAGS="$f" timeout 120 python bench.py --integrator stepham --system $s --batch 65536 --steps 20 --warmup 3 --no-cpu-baseline --no-isa 2>> $O/r05o.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(json.dumps({'system': '$s', 'flags': '$f', 'rep': $rep, 'calls_per_s': d['value'], 'kernel_ms': d['roofline']['kernel_ms'], 'mean_substeps': d['mean_substeps']}))" >> $O/r05o_swap_last_ab.jsonl
  done
done
done
cat $O/r05o_swap_last_ab.jsonl
timeout 400 python -m pytest tests/test_gpu_configs.py tests/test_gpu_wave.py -m gpu -q -x -k "adaptive or parked or c5_default or stepper" 2>&1 | tail -3
