#!/bin/bash
# GPU box, round 3, last pass (after the parked adaptive stepper and the measured dispatch thresholds): the whole GPU suite,
# smoke(), one bench line per config on ONE box, the reference's own stepper on every config system, the chain16 and
# chain32 adaptive-kernel profiles.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
export HAMK_TEST_RECORD=$PWD/gpurun_out/r03_gpu_test_record_f.jsonl
rm -f $HAMK_TEST_RECORD
timeout 2100 python -m pytest tests -m gpu -q > gpurun_out/gputest_f.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest_f.log
tail -6 gpurun_out/gputest_f.log
unset HAMK_TEST_RECORD
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_f.log 2>&1; tail -2 gpurun_out/smoke_f.log
for sys in doublePendulum twoBody spring threeBodyPolar chain8 chain16 chain32; do
  timeout 600 python bench.py --system $sys --steps 20 --warmup 5 > gpurun_out/bench_${sys}_f.json 2> gpurun_out/bench_${sys}_f.err
  head -c 200 gpurun_out/bench_${sys}_f.json; echo
done
for sys in doublePendulum twoBody spring threeBodyPolar chain8 chain16; do
  timeout 300 python bench.py --integrator stepham --system $sys --steps 20 --warmup 3 > gpurun_out/bench_stepham_${sys}_f.json 2> gpurun_out/bench_stepham_${sys}_f.err
  head -c 220 gpurun_out/bench_stepham_${sys}_f.json; echo
done
timeout 300 python bench.py --integrator stepham --system chain32 --batch 16384 --dt 0.02 --steps 10 --warmup 2 > gpurun_out/bench_stepham_chain32_f.json 2> gpurun_out/bench_stepham_chain32_f.err
head -c 220 gpurun_out/bench_stepham_chain32_f.json; echo
timeout 600 bash scripts/profile_stepham.sh r03i chain16 > /dev/null 2>&1
timeout 600 bash scripts/profile_stepham.sh r03i chain32 --batch 16384 --dt 0.02 > /dev/null 2>&1
ls gpurun_out/prof_r03i_chain16_stepham gpurun_out/prof_r03i_chain32_stepham
