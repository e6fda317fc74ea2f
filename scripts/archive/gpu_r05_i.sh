#!/bin/bash
# GPU box, round 5, pass I: the rotations / table evaluation in Horner + two-FMA form (HAMK_ROTATE_HORNER=1, the new default) against
# the power-basis form (=0), same box, back to back: RK4 lines of the systems that rotate (doublePendulum, pendulum) and of the ones
# that only take the table's two-FMA combine (twoBody, spring, threeBodyPolar, chain8), stepHam for two of them.
set -u
export HAMK_TEST_OVERRIDES=1
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export HAMK_CACHE_DIR=$PWD/.hamk_cache
O=gpurun_out; mkdir -p $O
: > $O/r05i_horner_ab.jsonl
for rep in 1 2; do
for s in doublePendulum pendulum twoBody spring threeBodyPolar chain8; do
  for f in "-DHAMK_ROTATE_HORNER=0" ""; do
    HAMK_HIPRTC_FLAGS="$f" timeout 120 python bench.py --system $s --steps 10 --warmup 3 --no-cpu-baseline --no-isa 2>> $O/r05i.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(json.dumps({'what': 'rk4', 'system': '$s', 'flags': '$f', 'rep': $rep, 'steps_per_s': d['value'], 'kernel_ms': d['roofline']['kernel_ms'], 'sclk_mhz': d['roofline']['fp64'].get('sclk_mhz_during_timed_region')}))" >> $O/r05i_horner_ab.jsonl
  done
done
done
for s in doublePendulum chain8; do
  for f in "-DHAMK_ROTATE_HORNER=0" ""; do
    HAMK_HIPRTC_FLAGS="$f" timeout 120 python bench.py --integrator stepham --system $s --steps 10 --warmup 3 --no-cpu-baseline --no-isa 2>> $O/r05i.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(json.dumps({'what': 'stepham', 'system': '$s', 'flags': '$f', 'calls_per_s': d['value']}))" >> $O/r05i_horner_ab.jsonl
  done
done
cat $O/r05i_horner_ab.jsonl | cut -c1-200
