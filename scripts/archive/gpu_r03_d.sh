#!/bin/bash
# GPU box, round 3, fourth pass: the quad module's new kernels (momenta, underlyingPos, keC, the adaptive stepper) on the
# GPU, random systems on the quad path, and the A/B measurements of scripts/quad_ab.py.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
export HAMK_TEST_RECORD=$PWD/gpurun_out/r03_gpu_test_record_d.jsonl
rm -f $HAMK_TEST_RECORD
timeout 1500 python -m pytest tests/test_gpu_wave.py tests/test_gpu_random_systems.py tests/test_gpu_configs.py -m gpu -q -k "quad or dense or other_code_paths or small_shard or evolveham or adaptive or c5_default or iterate" > gpurun_out/gputest_d.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest_d.log
tail -8 gpurun_out/gputest_d.log
timeout 1200 python scripts/quad_ab.py > gpurun_out/r03_quad_ab.jsonl 2> gpurun_out/quad_ab.err
cat gpurun_out/r03_quad_ab.jsonl
tail -3 gpurun_out/quad_ab.err
