#!/bin/bash
# GPU box, round 4, pass A: the whole GPU suite on the new code (pivoted solves, chain fixtures, C2 at 1000 steps, shard
# invariance, device sampler, bench.py's RCCL path), smoke(), the driver's bench line, one line per config, the reference's
# own stepper on the C4 / C5 systems (the starting point of this round's adaptive-stepper work).
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
export HAMK_TEST_RECORD=$PWD/gpurun_out/r04_gpu_test_record_a.jsonl
rm -f $HAMK_TEST_RECORD
timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/gputest_r04a.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest_r04a.log
tail -25 gpurun_out/gputest_r04a.log
unset HAMK_TEST_RECORD
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_r04a.log 2>&1; tail -2 gpurun_out/smoke_r04a.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r04a_doublePendulum.json 2> gpurun_out/bench_r04a_doublePendulum.err
head -c 300 gpurun_out/bench_r04a_doublePendulum.json; echo
for sys in threeBodyPolar chain8 chain16 chain32; do
  timeout 600 python bench.py --system $sys --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_r04a_${sys}.json 2> gpurun_out/bench_r04a_${sys}.err
  head -c 200 gpurun_out/bench_r04a_${sys}.json; echo
done
timeout 900 python scripts/rkf_prefetch_ab.py > gpurun_out/r04_rkf_prefetch_ab.jsonl 2> gpurun_out/r04_rkf_prefetch_ab.err
grep -c stepham gpurun_out/r04_rkf_prefetch_ab.jsonl; tail -3 gpurun_out/r04_rkf_prefetch_ab.err
for sys in threeBodyPolar chain8 chain16; do
  timeout 300 python bench.py --integrator stepham --system $sys --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r04a_stepham_${sys}.json 2> gpurun_out/bench_r04a_stepham_${sys}.err
  head -c 220 gpurun_out/bench_r04a_stepham_${sys}.json; echo
done
