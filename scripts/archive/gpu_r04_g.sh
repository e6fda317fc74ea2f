#!/bin/bash
# GPU box, round 4, pass G: A/B on ONE box of (a) the three factorisation orders of the four-lane kernels (HAMK_QUAD_LEFT 1 / 2), the
# arithmetic row pick (HAMK_QUAD_MSEL) and (b) LLVM's max-ILP scheduling strategy for the kernels that run one wavefront per SIMD.
# Every variant's code object is pre-compiled into .hamk_cache (the flags are part of the cache key).
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
OUT=gpurun_out/r04g_ab.jsonl
rm -f $OUT
ILP="-mllvm -amdgpu-sched-strategy=max-ilp"
run() {   # system, flags, extra bench args
  local sys=$1 fl=$2; shift 2
  if [ -n "$fl" ]; then export HAMK_HIPRTC_FLAGS="$fl"; else unset HAMK_HIPRTC_FLAGS; fi
  timeout 300 python bench.py --system $sys --no-cpu-baseline --no-isa "$@" 2>> gpurun_out/r04g_ab.err | tail -1 | SYS=$sys FL="$fl" ARGS="$*" python -c "
import sys, json, os
try:
    d = json.loads(sys.stdin.read())
    print(json.dumps({'system': os.environ['SYS'], 'flags': os.environ['FL'], 'args': os.environ['ARGS'], 'value': d['value'], 'ms_per_step': d['ms_per_step']}))
except Exception as e:
    print(json.dumps({'system': os.environ['SYS'], 'flags': os.environ['FL'], 'args': os.environ['ARGS'], 'error': repr(e)}))
" >> $OUT
  tail -1 $OUT
}
for rep in 1 2; do
for fl in "-DHAMK_QUAD_LEFT=1" "-DHAMK_QUAD_LEFT=2" "-DHAMK_QUAD_LEFT=1 $ILP" "-DHAMK_QUAD_LEFT=2 $ILP" "-DHAMK_QUAD_LEFT=1 -DHAMK_QUAD_MSEL=0" "-DHAMK_QUAD_LEFT=2 -DHAMK_QUAD_MSEL=0"; do
  run chain32 "$fl" --batch 65536 --steps 10 --warmup 3
done
done
for fl in "-DHAMK_QUAD_LEFT=1" "-DHAMK_QUAD_LEFT=2" "-DHAMK_QUAD_LEFT=1 $ILP" "-DHAMK_QUAD_LEFT=2 $ILP" "-DHAMK_QUAD_LEFT=1 -DHAMK_QUAD_MSEL=0" "-DHAMK_QUAD_LEFT=2 -DHAMK_QUAD_MSEL=0"; do
  run chain24 "$fl" --batch 65536 --steps 10 --warmup 3
done
for sys in chain16 chain12 chain8 threeBodyPolar doublePendulum; do
  for fl in "" "$ILP"; do
    BA=; [ $sys = chain12 ] && BA="--batch 65536"
    run $sys "$fl" --steps 10 --warmup 3 $BA
    run $sys "$fl" --integrator stepham --steps 20 --warmup 3 $BA
  done
done
wc -l $OUT
