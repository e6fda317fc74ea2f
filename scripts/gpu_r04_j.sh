#!/bin/bash
export HAMK_TEST_OVERRIDES=1   # the HAMK_* variables below are test overrides: read only when asked for
# GPU box, round 4, pass J (the final tree of the round: pass I without the table-gather bursts, which did not pay), most important first:
# the whole GPU suite, smoke(), one bench line per BASELINE config on ONE box, the reference's own stepper, bench.py's RCCL path,
# rocprofv3 stats + PMC for the headline kernel, chain32 and the adaptive stepper.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
export HAMK_TEST_RECORD=$PWD/gpurun_out/r04_gpu_test_record.jsonl
rm -f $HAMK_TEST_RECORD
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/gputest_r04j.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest_r04j.log
tail -16 gpurun_out/gputest_r04j.log
unset HAMK_TEST_RECORD
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_r04j.log 2>&1; tail -1 gpurun_out/smoke_r04j.log
rm -f gpurun_out/r04j_bench_configs.jsonl gpurun_out/r04j_bench_stepham.jsonl
for sys in doublePendulum chain32 chain16 chain8 threeBodyPolar twoBody spring; do
  CB=--no-cpu-baseline; [ $sys = doublePendulum ] && CB=
  timeout 400 python bench.py --system $sys --steps 20 --warmup 5 $CB 2> gpurun_out/bench_r04j_${sys}.err | tail -1 >> gpurun_out/r04j_bench_configs.jsonl
  tail -1 gpurun_out/r04j_bench_configs.jsonl | head -c 150; echo
done
for sys in threeBodyPolar chain16 chain8 doublePendulum twoBody spring; do
  timeout 300 python bench.py --integrator stepham --system $sys --steps 20 --warmup 3 2> gpurun_out/bench_r04j_stepham_${sys}.err | tail -1 >> gpurun_out/r04j_bench_stepham.jsonl
  tail -1 gpurun_out/r04j_bench_stepham.jsonl | head -c 190; echo
done
timeout 300 python bench.py --integrator stepham --system chain32 --batch 16384 --dt 0.02 --steps 10 --warmup 2 2>> gpurun_out/bench_r04j_stepham_chain32.err | tail -1 >> gpurun_out/r04j_bench_stepham.jsonl
MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --force-dist --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/bench_r04j_dist.err | grep "^{" | tail -1 > gpurun_out/r04j_bench_force_dist.json
head -c 160 gpurun_out/r04j_bench_force_dist.json; echo
unset HAMK_HIPRTC_FLAGS
timeout 400 bash scripts/profile.sh r04j doublePendulum > /dev/null 2>&1
timeout 400 bash scripts/profile.sh r04j chain32 > /dev/null 2>&1
timeout 400 bash scripts/profile_stepham.sh r04j chain8 > /dev/null 2>&1
timeout 400 bash scripts/profile_stepham.sh r04j threeBodyPolar > /dev/null 2>&1
timeout 400 bash scripts/profile_stepham.sh r04j chain16 > /dev/null 2>&1
ls gpurun_out | grep prof_r04j
