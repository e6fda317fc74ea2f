#!/bin/bash
# GPU box, round 6, pass A: the new tests (by-hand fixtures, bench.py --gpus self-launch), smoke, and this box's baseline lines of
# the kernels the round works on (C2 headline, chain16 / chain32 RK4, chain8 / chain16 stepHam, dense32 / chain64).
set -u
export HAMK_TEST_OVERRIDES=1
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export HAMK_CACHE_DIR=$PWD/.hamk_cache
O=gpurun_out; mkdir -p $O; T=r06a
timeout 900 python -m pytest tests -m gpu -q -k "by_hand or gpus_flag or two_ranks or rccl_path or golden_points" > $O/gputest_${T}.log 2>&1; echo "pytest rc=$?" >> $O/gputest_${T}.log
tail -5 $O/gputest_${T}.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_${T}.log 2>&1; tail -1 $O/smoke_${T}.log
rm -f $O/${T}_bench.jsonl
timeout 400 python bench.py --steps 20 --warmup 5 --cpu-seconds 5 2> $O/bench_${T}.err | tail -1 >> $O/${T}_bench.jsonl
for sys in chain16 chain32 chain8 threeBodyPolar; do
  timeout 300 python bench.py --system $sys --steps 20 --warmup 5 --no-cpu-baseline 2>> $O/bench_${T}.err | tail -1 >> $O/${T}_bench.jsonl
done
for sys in chain8 chain16; do
  timeout 300 python bench.py --integrator stepham --system $sys --steps 20 --warmup 3 --no-cpu-baseline 2>> $O/bench_${T}.err | tail -1 >> $O/${T}_bench.jsonl
done
for sys in dense32 chain64; do
  timeout 300 python bench.py --system $sys --batch 16384 --rk4-per-step 20 --steps 10 --warmup 2 --no-cpu-baseline 2>> $O/bench_${T}.err | tail -1 >> $O/${T}_bench.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r06a_bench.jsonl"):
    if l.startswith("{"):
        r = json.loads(l); print(r["config"]["workload"][:40], r.get("metric")[:20], "%.4g" % r["value"], r.get("roofline", {}).get("kernel_ms"))
PY
