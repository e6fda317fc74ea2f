#!/bin/bash
# GPU box, round 5, pass A (measure before code): where the parked adaptive stepper's cycles go (chain8, chain16), the starting
# numbers of the wave-cooperative kernels on this round's tree (dense24/32, chain48/64), the pair-row LDS layout against the
# shipped one for the RK4 kernel that parks its state, chain32 on this box.  Everything is pre-compiled (scripts/gpu_r05_a.sh prebuild).
set -u
export HAMK_TEST_OVERRIDES=1   # HAMK_SELFCHECK / HAMK_HIPRTC_FLAGS below are test overrides: read only when asked for
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export HAMK_CACHE_DIR=$PWD/.hamk_cache
CAND="--candidate=fixed5:-DHAMK_PROBE_FIXED=5 --candidate=fixed5-alias:-DHAMK_PROBE_FIXED=5_-DHAMK_PROBE_ALIAS_ROWS --candidate=fixed5-pair:-DHAMK_PROBE_FIXED=5_-DHAMK_PAIR_ROWS=1"
if [ "${1:-}" = "prebuild" ]; then
  python scripts/rkf_phase_probe.py --compile-only $CAND
  for s in dense24 dense32 chain48 chain64 chain32; do python -c "
import sys; sys.path.insert(0,'.')
from hamilton_amd import api, examples
s=api.system_from_spec(examples.get('$s')); print('$s', s.code_size)"; done
  HAMK_HIPRTC_FLAGS=-DHAMK_PAIR_ROWS=1 python -c "
import sys; sys.path.insert(0,'.')
from hamilton_amd import api, examples
for n in ('chain14','chain16'):
    s=api.system_from_spec(examples.get(n)); print(n, 'pair', s.code_size)"
  python -c "
import sys; sys.path.insert(0,'.')
from hamilton_amd import api, examples
s=api.system_from_spec(examples.get('chain14')); print('chain14', s.code_size)"
  exit 0
fi
mkdir -p gpurun_out
O=gpurun_out
timeout 400 python scripts/rkf_phase_probe.py $CAND > $O/r05a_rkf_phase_probe.jsonl 2> $O/r05a_rkf_phase_probe.err
tail -3 $O/r05a_rkf_phase_probe.err
BENCH="python bench.py --no-cpu-baseline --no-isa --steps 6 --warmup 2"
: > $O/r05a_wave_start.jsonl
for s in dense24 dense32 chain48 chain64; do
  timeout 200 $BENCH --system $s --batch 16384 --rk4-per-step 20 2>> $O/r05a_wave_start.err | tail -1 >> $O/r05a_wave_start.jsonl
done
: > $O/r05a_pair_rows_ab.jsonl
for s in chain14 chain16; do
  for f in "" "-DHAMK_PAIR_ROWS=1"; do
    HAMK_HIPRTC_FLAGS="$f" timeout 120 $BENCH --system $s --batch 65536 --rk4-per-step 200 2>> $O/r05a_pair_rows_ab.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(json.dumps({'system': '$s', 'flags': '$f', 'steps_per_s': d['value'], 'kernel_ms': d['roofline']['kernel_ms']}))" >> $O/r05a_pair_rows_ab.jsonl
  done
done
timeout 120 $BENCH --system chain32 2>/dev/null | tail -1 > $O/r05a_chain32.json
# PMC for the wave kernels' starting point (LDS unit, matrix cores, wait split)
bash scripts/profile.sh r05a dense32 --batch 16384 --rk4-per-step 20 > /dev/null 2>&1
python scripts/summarize_profile.py r05a dense32 > $O/r05a_summarize.log 2>&1
ls $O | head -50
