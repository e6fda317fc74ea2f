#!/bin/bash
# GPU box, round 3, first pass: the GPU suite, one bench line per config, quad vs wave on chain32, throughput vs B.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
export HAMK_TEST_RECORD=$PWD/gpurun_out/r03_gpu_test_record.jsonl
rm -f $HAMK_TEST_RECORD
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest.log
tail -25 gpurun_out/gputest.log
for sys in doublePendulum chain8 chain16 chain32 twoBody threeBodyPolar spring; do
  timeout 600 python bench.py --system $sys --steps 20 --warmup 5 > gpurun_out/bench_$sys.json 2> gpurun_out/bench_$sys.err
  head -c 400 gpurun_out/bench_$sys.json; echo
done
HAMK_QUAD=0 timeout 600 python bench.py --system chain32 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_chain32_wave.json 2> gpurun_out/bench_chain32_wave.err
HAMK_RK4_PARK=0 HAMK_K_REASSOC=0 timeout 600 python bench.py --system chain16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_chain16_r02.json 2> gpurun_out/bench_chain16_r02.err
HAMK_RK4_PARK=0 timeout 600 python bench.py --system chain16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_chain16_nopark.json 2> gpurun_out/bench_chain16_nopark.err
for f in chain32_wave chain16_r02 chain16_nopark; do head -c 300 gpurun_out/bench_$f.json; echo; done
timeout 900 python scripts/sweep_batch.py --out gpurun_out/r03_throughput_vs_B.jsonl > gpurun_out/sweep.log 2>&1
tail -5 gpurun_out/sweep.log
timeout 300 python bench.py --integrator stepham --steps 20 --warmup 3 > gpurun_out/bench_stepham_dp.json 2> gpurun_out/bench_stepham_dp.err
head -c 600 gpurun_out/bench_stepham_dp.json; echo
