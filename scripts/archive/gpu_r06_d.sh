#!/bin/bash
# GPU box, round 6, pass D: dense maps on the four-lane kernels after the shared products were unshared (one process per system, each under
# its own timeout), the C4 line after the fused 1/sqrt, the stepper lines after the fold
set -u
export HAMK_TEST_OVERRIDES=1
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export HAMK_CACHE_DIR=$PWD/.hamk_cache
O=gpurun_out; mkdir -p $O; T=r06d
rm -f $O/r06_dense_quad_ab.jsonl
for sys in dense18 dense24 denseD24 denseD32 dense32; do
  timeout 240 python scripts/dense_quad_ab.py $sys >> $O/r06_dense_quad_ab.jsonl 2>> $O/r06_dense_quad_ab.err; echo "$sys rc=$?"
done
rm -f $O/${T}_bench.jsonl
for sys in threeBodyPolar twoBody; do
  timeout 300 python bench.py --system $sys --steps 20 --warmup 5 --cpu-seconds 4 2>> $O/bench_${T}.err | tail -n 1 >> $O/${T}_bench.jsonl
done
for sys in chain16 chain14 threeBodyPolar; do
  timeout 300 python bench.py --integrator stepham --system $sys --steps 20 --warmup 3 --no-cpu-baseline 2>> $O/bench_${T}.err | tail -n 1 >> $O/${T}_bench.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r06_dense_quad_ab.jsonl"):
    if l.startswith("{"):
        r = json.loads(l); print(r["system"], r["mapping"], r["B"], "%.4g" % r["rk4_steps_per_s"], r["flagged"], "%.2g %.2g" % (r["hameqs_rel_err_vs_oracle"], r["one_step_rel_err_vs_oracle"]))
for l in open("gpurun_out/r06d_bench.jsonl"):
    if l.startswith("{"):
        r = json.loads(l); print(r["config"]["workload"][:40], r.get("metric")[:20], "%.4g" % r["value"], r.get("roofline", {}).get("kernel_ms"), r.get("parity", {}).get("max_abs_dphase_1_step"))
PY
