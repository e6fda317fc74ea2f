#!/bin/bash
# GPU box, round 3, final pass: the whole GPU suite, smoke(), one bench line per config on ONE box, the chain32 profile
# (stats + PMC incl. LDS) for the shipped factorisation, stepHam lines on the quad kernels.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
export HAMK_TEST_RECORD=$PWD/gpurun_out/r03_gpu_test_record_e.jsonl
rm -f $HAMK_TEST_RECORD
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/gputest_e.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest_e.log
tail -6 gpurun_out/gputest_e.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_e.log 2>&1; tail -2 gpurun_out/smoke_e.log
for sys in doublePendulum twoBody spring threeBodyPolar chain8 chain16 chain32; do
  timeout 600 python bench.py --system $sys --steps 20 --warmup 5 > gpurun_out/bench_${sys}_e.json 2> gpurun_out/bench_${sys}_e.err
  head -c 200 gpurun_out/bench_${sys}_e.json; echo
done
HAMK_QUAD=0 timeout 600 python bench.py --system chain32 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_chain32_wave_e.json 2> gpurun_out/bench_chain32_wave_e.err
head -c 200 gpurun_out/bench_chain32_wave_e.json; echo
timeout 900 bash scripts/profile.sh r03e chain32 > gpurun_out/profile_chain32_e.log 2>&1
timeout 300 python bench.py --integrator stepham --system chain32 --batch 16384 --dt 0.02 --steps 10 --warmup 2 > gpurun_out/bench_stepham_chain32_e.json 2> gpurun_out/bench_stepham_chain32_e.err
timeout 300 python bench.py --integrator stepham --steps 20 --warmup 3 > gpurun_out/bench_stepham_dp_e.json 2> gpurun_out/bench_stepham_dp_e.err
head -c 300 gpurun_out/bench_stepham_chain32_e.json; echo
timeout 600 python scripts/sweep_batch.py --out gpurun_out/r03_throughput_vs_B_quad_e.jsonl --systems chain32,chain16 --mappings quad > gpurun_out/sweep_e.log 2>&1
tail -2 gpurun_out/sweep_e.log
