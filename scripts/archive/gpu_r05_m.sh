#!/bin/bash
# GPU box, round 5, pass M: with the rotations at 14-16 instructions (Horner form) the rule "table for every evaluation unless 1-4
# sites and a short right-hand side" is measured again: RK4 lines with HAMK_TRIG_LUT=2 (one table evaluation per step + rotations)
# against the default, same box back to back.
set -u
export HAMK_TEST_OVERRIDES=1
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export HAMK_CACHE_DIR=$PWD/.hamk_cache
O=gpurun_out; mkdir -p $O
: > $O/r05m_trig_rule_ab.jsonl
for s in twoBody spring threeBodyPolar chain4 chain6 chain8 chain10 chain12 chain14; do
  B=""; case $s in chain4|chain6|chain10|chain12|chain14) B="--batch 65536 --rk4-per-step 200";; esac
  for f in "" "2"; do
    if [ -z "$f" ]; then unset HAMK_TRIG_LUT; else export HAMK_TRIG_LUT=$f; fi
    timeout 120 python bench.py --system $s $B --steps 8 --warmup 2 --no-cpu-baseline --no-isa 2>> $O/r05m.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(json.dumps({'system': '$s', 'HAMK_TRIG_LUT': '$f' or 'default', 'steps_per_s': d['value'], 'kernel_ms': d['roofline']['kernel_ms']}))" >> $O/r05m_trig_rule_ab.jsonl
  done
done
unset HAMK_TRIG_LUT
cat $O/r05m_trig_rule_ab.jsonl
