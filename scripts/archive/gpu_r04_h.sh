#!/bin/bash
# GPU box, round 4, pass H: A/B on ONE box of the burst loads of the four-lane kernels' sweeps (HAMK_QUAD_BURST), the second look at the LDS
# rows in the reverse pass (HAMK_QUAD_RELOAD), the look-ahead order and the arithmetic row pick on top of the bursts; then the PMC
# passes that say where a wavefront's cycles go (SQ_WAIT_ANY / SQ_ACTIVE_INST_ANY) for the default build.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
OUT=gpurun_out/r04h_ab.jsonl
rm -f $OUT
run() {   # system, flags, extra bench args
  local sys=$1 fl=$2; shift 2
  if [ -n "$fl" ]; then export HAMK_HIPRTC_FLAGS="$fl"; else unset HAMK_HIPRTC_FLAGS; fi
  timeout 300 python bench.py --system $sys --no-cpu-baseline --no-isa "$@" 2>> gpurun_out/r04h_ab.err | tail -1 | SYS=$sys FL="$fl" ARGS="$*" python -c "
import sys, json, os
try:
    d = json.loads(sys.stdin.read())
    print(json.dumps({'system': os.environ['SYS'], 'flags': os.environ['FL'], 'args': os.environ['ARGS'], 'value': d['value'], 'ms_per_step': d['ms_per_step']}))
except Exception as e:
    print(json.dumps({'system': os.environ['SYS'], 'flags': os.environ['FL'], 'args': os.environ['ARGS'], 'error': repr(e)}))
" >> $OUT
  tail -1 $OUT
}
for sys in chain32 chain24 chain17; do
for rep in 1 2; do
for fl in "-DHAMK_QUAD_BURST=0" "-DHAMK_QUAD_BURST=1" "-DHAMK_QUAD_BURST=1 -DHAMK_QUAD_RELOAD=0" "-DHAMK_QUAD_BURST=1 -DHAMK_QUAD_LEFT=2" "-DHAMK_QUAD_BURST=1 -DHAMK_QUAD_MSEL=0"; do
  run $sys "$fl" --batch 65536 --steps 10 --warmup 3
done
done
done
unset HAMK_HIPRTC_FLAGS
export TMPDIR=/tmp HAMK_SELFCHECK=0
R=$PWD; P=$R/gpurun_out/prof_r04h_chain32; mkdir -p $P; cd /tmp
B="python $R/bench.py --system chain32 --no-cpu-baseline --no-isa --steps 3 --warmup 1"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $P/pmc_wait -o r04h -- $B > $P/pmc_wait.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS --output-format csv -d $P/pmc_ifetch -o r04h -- $B > $P/pmc_ifetch.log 2>&1
ls $P
