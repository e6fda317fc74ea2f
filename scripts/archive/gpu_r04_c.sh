#!/bin/bash
# GPU box, round 4, pass C (the final tree of the round): the whole GPU suite (every failure listed, not just the first), smoke(), one
# bench line per BASELINE config on ONE box, the reference's own stepper on every config system, bench.py's RCCL path, and
# the rocprofv3 passes (stats + PMC) the profiles/ summaries are made from.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
export HAMK_TEST_RECORD=$PWD/gpurun_out/r04_gpu_test_record.jsonl
rm -f $HAMK_TEST_RECORD
timeout 2400 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/gputest_r04c.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest_r04c.log
tail -40 gpurun_out/gputest_r04c.log
unset HAMK_TEST_RECORD
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_r04c.log 2>&1; tail -2 gpurun_out/smoke_r04c.log
rm -f gpurun_out/r04_bench_configs.jsonl gpurun_out/r04_bench_stepham.jsonl
for sys in doublePendulum twoBody spring threeBodyPolar chain8 chain16 chain32; do
  timeout 600 python bench.py --system $sys --steps 20 --warmup 5 2> gpurun_out/bench_r04c_${sys}.err | tail -1 >> gpurun_out/r04_bench_configs.jsonl
  tail -1 gpurun_out/r04_bench_configs.jsonl | head -c 160; echo
done
for sys in doublePendulum twoBody spring threeBodyPolar chain8 chain16; do
  timeout 300 python bench.py --integrator stepham --system $sys --steps 20 --warmup 3 2> gpurun_out/bench_r04c_stepham_${sys}.err | tail -1 >> gpurun_out/r04_bench_stepham.jsonl
  tail -1 gpurun_out/r04_bench_stepham.jsonl | head -c 200; echo
done
timeout 300 python bench.py --integrator stepham --system chain32 --batch 16384 --dt 0.02 --steps 10 --warmup 2 2>> gpurun_out/bench_r04c_stepham_chain32.err | tail -1 >> gpurun_out/r04_bench_stepham.jsonl
MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --force-dist --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/bench_r04c_dist.err | grep "^{" | tail -1 > gpurun_out/r04_bench_force_dist.json
head -c 200 gpurun_out/r04_bench_force_dist.json; echo
timeout 600 bash scripts/profile.sh r04c doublePendulum > /dev/null 2>&1
timeout 900 bash scripts/profile.sh r04c chain32 > /dev/null 2>&1
ls gpurun_out | grep prof_r04c
