#!/bin/bash
# GPU box, round 6, pass C: stepper A/B (fold, frcp in two forms), the new GPU tests (dense maps on the four-lane kernels, cold path of the
# cooperative mappings, domain edge, reference-document comparison), smoke
set -u
export HAMK_TEST_OVERRIDES=1
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export HAMK_CACHE_DIR=$PWD/.hamk_cache
O=gpurun_out; mkdir -p $O; T=r06c
timeout 300 python scripts/rkf_fold_ab.py > $O/r06_rkf_fold_ab.jsonl 2> $O/r06_rkf_fold_ab.err; echo "rkf_fold_ab rc=$?"
timeout 700 python -m pytest tests -m gpu -q -x --durations=6 -k "quad_path_vs_oracle or dense_jacobians or cold_path or edge_of_their_domain or reference_haskell or heavy_tapes or by_hand" > $O/gputest_${T}.log 2>&1; echo "pytest rc=$?" >> $O/gputest_${T}.log
tail -n 16 $O/gputest_${T}.log
python - <<'PY'
import json
for l in open("gpurun_out/r06_rkf_fold_ab.jsonl"):
    if l.startswith("{"):
        r = json.loads(l); print(r["system"], r["variant"], r["dt_mult"], "%.4g" % r["calls_per_s"], r.get("identical_substep_counts_frac"), r.get("max_abs_diff_to_round5"))
PY
