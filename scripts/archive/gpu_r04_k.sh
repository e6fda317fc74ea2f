#!/bin/bash
# GPU box, round 4, pass K (bench lines only: the GPU minutes of the round are spent): A/B on ONE box of the burst row loads of the
# parked adaptive stepper (HAMK_RKF_BURST) -- every line carries the bench's own check against the oracle (identical sub-step
# counts, max |dphase|).  Both code objects of every system are pre-compiled into .hamk_cache.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
OUT=gpurun_out/r04k_ab.jsonl
rm -f $OUT
run() {
  local sys=$1 fl=$2; shift 2
  if [ -n "$fl" ]; then export HAMK_HIPRTC_FLAGS="$fl"; else unset HAMK_HIPRTC_FLAGS; fi
  timeout 120 python bench.py --system $sys --no-cpu-baseline --no-isa "$@" 2>> gpurun_out/r04k_ab.err | tail -1 | SYS=$sys FL="$fl" ARGS="$*" python -c "
import sys, json, os
try:
    d = json.loads(sys.stdin.read()); p = d.get('parity', {})
    print(json.dumps({'system': os.environ['SYS'], 'flags': os.environ['FL'], 'args': os.environ['ARGS'], 'value': d['value'], 'ms_per_step': d['ms_per_step'],
                      'identical_substep_counts_frac': p.get('identical_substep_counts_frac'), 'max_abs_dphase_lanes_with_identical_counts': p.get('max_abs_dphase_lanes_with_identical_counts'),
                      'status_flagged': d.get('status_flagged')}))
except Exception as e:
    print(json.dumps({'system': os.environ['SYS'], 'flags': os.environ['FL'], 'args': os.environ['ARGS'], 'error': repr(e)}))
" >> $OUT
  tail -1 $OUT | head -c 300; echo
}
for rep in 1 2; do
for sys in chain16 chain12 chain10 chain8; do
  for fl in "" "-DHAMK_RKF_BURST=0"; do
    run $sys "$fl" --batch 65536 --integrator stepham --steps 20 --warmup 3
  done
done
done
