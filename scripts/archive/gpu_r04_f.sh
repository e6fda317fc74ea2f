#!/bin/bash
# GPU box, round 4, pass F (the tree with the in-place left-looking Cholesky of the four-lane kernels, the re-associated jet gradients and
# the reloading reverse sweep), most important first: the whole GPU suite, smoke(), one bench line per BASELINE config on ONE box, the
# chain32 A/B of the two factorisation orders, the reference's own stepper, bench.py's RCCL path, rocprofv3 stats + PMC.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
export HAMK_TEST_RECORD=$PWD/gpurun_out/r04_gpu_test_record.jsonl
rm -f $HAMK_TEST_RECORD
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/gputest_r04f.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest_r04f.log
tail -25 gpurun_out/gputest_r04f.log
unset HAMK_TEST_RECORD
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_r04f.log 2>&1; tail -2 gpurun_out/smoke_r04f.log
rm -f gpurun_out/r04f_bench_configs.jsonl gpurun_out/r04f_bench_stepham.jsonl gpurun_out/r04f_chain32_ab.jsonl
for sys in doublePendulum chain32 chain16 chain8 threeBodyPolar twoBody spring; do
  CB=--no-cpu-baseline; [ $sys = doublePendulum ] && CB=
  timeout 400 python bench.py --system $sys --steps 20 --warmup 5 $CB 2> gpurun_out/bench_r04f_${sys}.err | tail -1 >> gpurun_out/r04f_bench_configs.jsonl
  tail -1 gpurun_out/r04f_bench_configs.jsonl | head -c 160; echo
done
# chain32: right-looking rank-4 panels against the left-looking order (same Cholesky, same storage), and chain24 / chain20
for v in 0 1; do
  for sys in chain32 chain24; do
    HAMK_HIPRTC_FLAGS="-DHAMK_QUAD_LEFT=$v" timeout 400 python bench.py --system $sys --batch 65536 --steps 10 --warmup 3 --no-cpu-baseline --no-isa 2>> gpurun_out/bench_r04f_ab.err | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(json.dumps({'system': '$sys', 'HAMK_QUAD_LEFT': $v, 'value': d['value'], 'ms_per_step': d['ms_per_step']}))" >> gpurun_out/r04f_chain32_ab.jsonl
  done
done
cat gpurun_out/r04f_chain32_ab.jsonl
for sys in threeBodyPolar chain16 chain8 doublePendulum twoBody spring; do
  timeout 300 python bench.py --integrator stepham --system $sys --steps 20 --warmup 3 2> gpurun_out/bench_r04f_stepham_${sys}.err | tail -1 >> gpurun_out/r04f_bench_stepham.jsonl
  tail -1 gpurun_out/r04f_bench_stepham.jsonl | head -c 200; echo
done
timeout 300 python bench.py --integrator stepham --system chain32 --batch 16384 --dt 0.02 --steps 10 --warmup 2 2>> gpurun_out/bench_r04f_stepham_chain32.err | tail -1 >> gpurun_out/r04f_bench_stepham.jsonl
MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --force-dist --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/bench_r04f_dist.err | grep "^{" | tail -1 > gpurun_out/r04f_bench_force_dist.json
head -c 200 gpurun_out/r04f_bench_force_dist.json; echo
timeout 400 bash scripts/profile.sh r04f chain32 > /dev/null 2>&1
timeout 400 bash scripts/profile.sh r04f doublePendulum > /dev/null 2>&1
timeout 400 bash scripts/profile_stepham.sh r04f threeBodyPolar > /dev/null 2>&1
timeout 400 bash scripts/profile_stepham.sh r04f chain16 > /dev/null 2>&1
ls gpurun_out | grep prof_r04f
