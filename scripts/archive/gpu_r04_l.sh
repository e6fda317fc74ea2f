#!/bin/bash
# GPU box, round 4, pass L (bench lines only, a slim snapshot: .hamk_cache stays behind, the four code objects needed are in
# .hamk_cache_ab): the sincos table against the direct evaluation for the lane kernels that park their RK4 state (n >= 14).
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
export HAMK_CACHE_DIR=$PWD/.hamk_cache_ab
OUT=gpurun_out/r04l_ab.jsonl
rm -f $OUT
for rep in 1 2; do
for sys in chain16 chain14; do
  for lut in "" 0; do
    for mode in "" "--integrator stepham"; do
      if [ -n "$lut" ]; then export HAMK_TRIG_LUT=$lut; else unset HAMK_TRIG_LUT; fi
      timeout 100 python bench.py --system $sys --batch 65536 --no-cpu-baseline --no-isa --steps 10 --warmup 3 $mode 2>> gpurun_out/r04l_ab.err | tail -1 | SYS=$sys LUT="$lut" MODE="$mode" python -c "
import sys, json, os
try:
    d = json.loads(sys.stdin.read())
    print(json.dumps({'system': os.environ['SYS'], 'HAMK_TRIG_LUT': os.environ['LUT'] or 'default', 'mode': os.environ['MODE'] or 'rk4', 'value': d['value'], 'ms_per_step': d['ms_per_step']}))
except Exception as e:
    print(json.dumps({'system': os.environ['SYS'], 'HAMK_TRIG_LUT': os.environ['LUT'], 'mode': os.environ['MODE'], 'error': repr(e)}))
" >> $OUT
      tail -1 $OUT
    done
  done
done
done
