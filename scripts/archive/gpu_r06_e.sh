#!/bin/bash
# GPU box, round 6, pass E: the tree with the symbolic mass matrix -- bench lines of every BASELINE config (RK4 + the reference's own stepper),
# then the whole GPU suite + smoke
set -u
export HAMK_TEST_OVERRIDES=1
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export HAMK_CACHE_DIR=$PWD/.hamk_cache
O=gpurun_out; mkdir -p $O; T=${TAG:-r06e}
PARTS="${@:-bench tests}"
if [[ " $PARTS " == *" bench "* ]]; then
  rm -f $O/${T}_bench_configs.jsonl $O/${T}_bench_stepham.jsonl
  for sys in doublePendulum twoBody spring threeBodyPolar chain8 chain16 chain32; do
    timeout 400 python bench.py --system $sys --steps 20 --warmup 5 --cpu-seconds 4 2> $O/bench_${T}_${sys}.err | tail -n 1 >> $O/${T}_bench_configs.jsonl
  done
  for sys in doublePendulum twoBody spring threeBodyPolar chain8 chain16; do
    timeout 300 python bench.py --integrator stepham --system $sys --steps 20 --warmup 3 2> $O/bench_${T}_stepham_${sys}.err | tail -n 1 >> $O/${T}_bench_stepham.jsonl
  done
  python - <<PY
import json
for f in ("$O/${T}_bench_configs.jsonl", "$O/${T}_bench_stepham.jsonl"):
    for l in open(f):
        if l.startswith("{"):
            r = json.loads(l); rf = r.get("roofline", {}); fp = rf.get("fp64", {})
            print(r["config"]["workload"][:36], "%.4g" % r["value"], "frac8d", rf.get("frac") and round(rf["frac"], 3), "valu", fp.get("valu_insts_per_wave_step"), "fp64frac", fp.get("frac_of_peak") and round(fp["frac_of_peak"], 3),
                  "parity1", r.get("parity", {}).get("max_abs_dphase_1_step"), "counts", r.get("parity", {}).get("identical_substep_counts_frac"), "flagged", r.get("status_flagged"))
PY
fi
if [[ " $PARTS " == *" tests "* ]]; then
  export HAMK_TEST_RECORD=$PWD/$O/${T}_gpu_test_record.jsonl
  rm -f $HAMK_TEST_RECORD
  timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/gputest_${T}.log 2>&1; echo "pytest rc=$?" >> $O/gputest_${T}.log
  tail -n 16 $O/gputest_${T}.log
  unset HAMK_TEST_RECORD
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_${T}.log 2>&1; tail -n 1 $O/smoke_${T}.log
fi
