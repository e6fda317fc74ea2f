#!/bin/bash
# GPU box, round 3, second pass: re-run of the tests touched since pass A, chain32 on the panel-factorising quad kernels,
# throughput vs B incl. the quad mapping, rocprofv3 stats + PMC (incl. LDS) for chain32 / chain16 / doublePendulum, and
# the reference's own stepper (hamk_rkf45_k) on three systems.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
export HAMK_TEST_RECORD=$PWD/gpurun_out/r03_gpu_test_record_b.jsonl
rm -f $HAMK_TEST_RECORD
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_wave.py tests/test_cpp_host.py -m gpu -q -k "full_size or quad or flat_and_panel or dense or cpp or c_client" > gpurun_out/gputest_b.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest_b.log
tail -8 gpurun_out/gputest_b.log
timeout 600 python bench.py --system chain32 --steps 20 --warmup 5 > gpurun_out/bench_chain32_b.json 2> gpurun_out/bench_chain32_b.err
head -c 300 gpurun_out/bench_chain32_b.json; echo
timeout 900 python scripts/sweep_batch.py --out gpurun_out/r03_throughput_vs_B_quad.jsonl --systems chain32,chain16,chain8,threeBodyPolar --mappings quad > gpurun_out/sweep_b.log 2>&1
tail -3 gpurun_out/sweep_b.log
for sys in chain32 chain16 doublePendulum; do
  timeout 900 bash scripts/profile.sh r03 $sys > gpurun_out/profile_$sys.log 2>&1
done
for sys in doublePendulum spring threeBodyPolar; do
  timeout 600 bash scripts/profile_stepham.sh r03 $sys > gpurun_out/profile_stepham_$sys.log 2>&1
done
timeout 300 python bench.py --integrator stepham --steps 20 --warmup 3 > gpurun_out/bench_stepham_dp.json 2> gpurun_out/bench_stepham_dp.err
timeout 300 python bench.py --integrator stepham --steps 10 --warmup 2 --calls-per-launch 32 --no-cpu-baseline > gpurun_out/bench_stepham_dp_x32.json 2> gpurun_out/bench_stepham_dp_x32.err
timeout 300 python bench.py --integrator stepham --system threeBodyPolar --steps 20 --warmup 3 > gpurun_out/bench_stepham_tbp.json 2> gpurun_out/bench_stepham_tbp.err
timeout 300 python bench.py --integrator stepham --system spring --steps 20 --warmup 3 > gpurun_out/bench_stepham_spring.json 2> gpurun_out/bench_stepham_spring.err
for f in stepham_dp stepham_dp_x32 stepham_tbp stepham_spring; do head -c 250 gpurun_out/bench_$f.json; echo; done
ls gpurun_out | head -80
