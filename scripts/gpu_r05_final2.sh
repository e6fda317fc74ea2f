#!/bin/bash
# GPU box, round 5: after the last device-header change (the stepper's last-stage LDS swap) -- the whole GPU suite again, the
# stepHam lines of all seven systems and the stepper's profiles, on the final tree.
set -u
export HAMK_TEST_OVERRIDES=1
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export HAMK_CACHE_DIR=$PWD/.hamk_cache
O=gpurun_out; mkdir -p $O
bash scripts/gpu_r05_final.sh tests
rm -f $O/r05_bench_stepham.jsonl
for sys in doublePendulum twoBody spring threeBodyPolar chain8 chain16; do
  timeout 300 python bench.py --integrator stepham --system $sys --steps 20 --warmup 3 2> $O/bench_r05_stepham_${sys}.err | tail -1 >> $O/r05_bench_stepham.jsonl
  tail -1 $O/r05_bench_stepham.jsonl | head -c 170; echo
done
timeout 300 python bench.py --integrator stepham --system chain32 --batch 16384 --dt 0.02 --steps 10 --warmup 2 2>> $O/bench_r05_stepham_chain32.err | tail -1 >> $O/r05_bench_stepham.jsonl
for sys in chain8 chain16; do timeout 400 bash scripts/profile_stepham.sh r05 $sys > /dev/null 2>&1; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_line_final.json; head -c 150 $O/r05_bench_line_final.json; echo
