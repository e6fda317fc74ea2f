#!/bin/bash
# GPU box, round 3, third pass: the whole GPU suite on the final code, left- vs right-looking quad factorisation on
# chain32, one bench line per config, profile of the chain32 winner.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
export HAMK_TEST_RECORD=$PWD/gpurun_out/r03_gpu_test_record_c.jsonl
rm -f $HAMK_TEST_RECORD
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/gputest_c.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest_c.log
tail -12 gpurun_out/gputest_c.log
timeout 600 python bench.py --system chain32 --steps 20 --warmup 5 > gpurun_out/bench_chain32_left.json 2> gpurun_out/bench_chain32_left.err
HAMK_HIPRTC_FLAGS="-DHAMK_QUAD_LEFT=0" timeout 600 python bench.py --system chain32 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_chain32_right.json 2> gpurun_out/bench_chain32_right.err
for f in chain32_left chain32_right; do head -c 260 gpurun_out/bench_$f.json; echo; done
timeout 900 python scripts/sweep_batch.py --out gpurun_out/r03_throughput_vs_B_quad_c.jsonl --systems chain32,chain16,chain24 --mappings quad > gpurun_out/sweep_c.log 2>&1
tail -3 gpurun_out/sweep_c.log
timeout 900 bash scripts/profile.sh r03c chain32 > gpurun_out/profile_chain32_c.log 2>&1
for sys in doublePendulum chain8 chain16 twoBody threeBodyPolar spring; do
  timeout 600 python bench.py --system $sys --steps 20 --warmup 5 > gpurun_out/bench_${sys}_c.json 2> gpurun_out/bench_${sys}_c.err
  head -c 200 gpurun_out/bench_${sys}_c.json; echo
done
