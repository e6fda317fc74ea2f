#!/bin/bash
# GPU box, round 4, pass I (the final tree of the round: in-place left-looking Cholesky, burst loads of the sweeps' LDS rows and of the
# sincos table gathers), most important first: the whole GPU suite, smoke(), one bench line per BASELINE config on ONE box, the
# reference's own stepper, bench.py's RCCL path, the A/B of the table-gather burst and of the look-ahead order, rocprofv3 stats + PMC.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
export HAMK_TEST_RECORD=$PWD/gpurun_out/r04_gpu_test_record.jsonl
rm -f $HAMK_TEST_RECORD
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/gputest_r04i.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest_r04i.log
tail -16 gpurun_out/gputest_r04i.log
unset HAMK_TEST_RECORD
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_r04i.log 2>&1; tail -1 gpurun_out/smoke_r04i.log
rm -f gpurun_out/r04i_bench_configs.jsonl gpurun_out/r04i_bench_stepham.jsonl gpurun_out/r04i_ab.jsonl
for sys in doublePendulum chain32 chain16 chain8 threeBodyPolar twoBody spring; do
  CB=--no-cpu-baseline; [ $sys = doublePendulum ] && CB=
  timeout 400 python bench.py --system $sys --steps 20 --warmup 5 $CB 2> gpurun_out/bench_r04i_${sys}.err | tail -1 >> gpurun_out/r04i_bench_configs.jsonl
  tail -1 gpurun_out/r04i_bench_configs.jsonl | head -c 150; echo
done
for sys in threeBodyPolar chain16 chain8 doublePendulum twoBody spring; do
  timeout 300 python bench.py --integrator stepham --system $sys --steps 20 --warmup 3 2> gpurun_out/bench_r04i_stepham_${sys}.err | tail -1 >> gpurun_out/r04i_bench_stepham.jsonl
  tail -1 gpurun_out/r04i_bench_stepham.jsonl | head -c 190; echo
done
timeout 300 python bench.py --integrator stepham --system chain32 --batch 16384 --dt 0.02 --steps 10 --warmup 2 2>> gpurun_out/bench_r04i_stepham_chain32.err | tail -1 >> gpurun_out/r04i_bench_stepham.jsonl
MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --force-dist --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/bench_r04i_dist.err | grep "^{" | tail -1 > gpurun_out/r04i_bench_force_dist.json
head -c 160 gpurun_out/r04i_bench_force_dist.json; echo
run() {   # system, flags, extra bench args
  local sys=$1 fl=$2; shift 2
  if [ -n "$fl" ]; then export HAMK_HIPRTC_FLAGS="$fl"; else unset HAMK_HIPRTC_FLAGS; fi
  timeout 300 python bench.py --system $sys --no-cpu-baseline --no-isa "$@" 2>> gpurun_out/r04i_ab.err | tail -1 | SYS=$sys FL="$fl" ARGS="$*" python -c "
import sys, json, os
try:
    d = json.loads(sys.stdin.read())
    print(json.dumps({'system': os.environ['SYS'], 'flags': os.environ['FL'], 'args': os.environ['ARGS'], 'value': d['value'], 'ms_per_step': d['ms_per_step']}))
except Exception as e:
    print(json.dumps({'system': os.environ['SYS'], 'flags': os.environ['FL'], 'args': os.environ['ARGS'], 'error': repr(e)}))
" >> gpurun_out/r04i_ab.jsonl
  tail -1 gpurun_out/r04i_ab.jsonl | head -c 230; echo
}
for sys in chain8 chain12 chain16; do
  for fl in "" "-DHAMK_LUT_BURST_MIN=99"; do
    run $sys "$fl" --batch 65536 --steps 10 --warmup 3
    run $sys "$fl" --batch 65536 --integrator stepham --steps 20 --warmup 3
  done
done
for sys in chain32 chain24 chain17; do
  for fl in "" "-DHAMK_QUAD_LEFT=2"; do
    run $sys "$fl" --batch 65536 --steps 10 --warmup 3
  done
done
run chain12 "" --batch 8192 --steps 10 --warmup 3
run chain16 "" --batch 8192 --steps 10 --warmup 3
unset HAMK_HIPRTC_FLAGS
timeout 400 bash scripts/profile.sh r04i doublePendulum > /dev/null 2>&1
timeout 400 bash scripts/profile.sh r04i chain32 > /dev/null 2>&1
timeout 400 bash scripts/profile.sh r04i chain8 > /dev/null 2>&1
timeout 400 bash scripts/profile.sh r04i chain16 > /dev/null 2>&1
timeout 400 bash scripts/profile_stepham.sh r04i chain8 > /dev/null 2>&1
timeout 400 bash scripts/profile_stepham.sh r04i threeBodyPolar > /dev/null 2>&1
ls gpurun_out | grep prof_r04i
