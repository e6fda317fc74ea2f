#!/bin/bash
# GPU box, round 6, pass B: A/B of the parked adaptive stepper (fold / frcp) and of dense maps on the four-lane kernels vs the wave kernels
set -u
export HAMK_TEST_OVERRIDES=1
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export HAMK_CACHE_DIR=$PWD/.hamk_cache
O=gpurun_out; mkdir -p $O
[ -z "${SKIP_RKF:-}" ] && timeout 900 python scripts/rkf_fold_ab.py > $O/r06_rkf_fold_ab.jsonl 2> $O/r06_rkf_fold_ab.err; echo "rkf_fold_ab rc=$?"
[ -z "${SKIP_DENSE:-}" ] && timeout 1200 python scripts/dense_quad_ab.py ${DENSE:-dense18 dense24} > $O/r06_dense_quad_ab.jsonl 2> $O/r06_dense_quad_ab.err; echo "dense_quad_ab rc=$?"
python - <<'PY'
import json
import os
for f in ("gpurun_out/r06_rkf_fold_ab.jsonl", "gpurun_out/r06_dense_quad_ab.jsonl"):
    for l in (open(f) if os.path.exists(f) else []):
        if not l.startswith("{"): continue
        r = json.loads(l)
        if r["what"] == "stepham":
            print(r["system"], r["variant"], r["dt_mult"], "%.4g" % r["calls_per_s"], r["mean_substeps"], r.get("identical_substep_counts_frac"), r.get("max_abs_diff_to_round5"))
        else:
            print(r["system"], r["mapping"], r["B"], "%.4g" % r["rk4_steps_per_s"], r["flagged"], "%.2g %.2g" % (r["hameqs_rel_err_vs_oracle"], r["one_step_rel_err_vs_oracle"]))
PY
tail -n 3 $O/r06_rkf_fold_ab.err $O/r06_dense_quad_ab.err
