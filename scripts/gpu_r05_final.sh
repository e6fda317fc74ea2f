#!/bin/bash
# GPU box, round 5, the final-tree evidence pass -- ONE box, everything with cpu_baseline + parity:
#   part "tests":  the whole GPU suite + smoke()
#   part "bench":  one RK4 line per BASELINE config (seven systems) WITH cpu_baseline and parity, the reference's own stepper on
#                  all seven, the wave-cooperative kernels (dense24/32, chain48/64), the RCCL path on one rank
#   part "prof":   rocprofv3 stats + PMC (HBM traffic -> profiles/pmc_traffic_*.json, wait split, LDS, matrix cores) for the
#                  headline kernel, the C4/C5 kernels, the stepper at n = 8, 16, and the wave kernels
# usage: scripts/gpu_r05_final.sh [tests] [bench] [prof]   (default: all three).  Everything is pre-compiled (scripts/warm_cache.py).
set -u
export HAMK_TEST_OVERRIDES=1
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export HAMK_CACHE_DIR=$PWD/.hamk_cache
O=gpurun_out
mkdir -p $O
PARTS="${@:-tests bench prof}"
T=r05
if [[ " $PARTS " == *" tests "* ]]; then
  export HAMK_TEST_RECORD=$PWD/$O/${T}_gpu_test_record.jsonl
  rm -f $HAMK_TEST_RECORD
  timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/gputest_${T}.log 2>&1; echo "pytest rc=$?" >> $O/gputest_${T}.log
  tail -14 $O/gputest_${T}.log
  unset HAMK_TEST_RECORD
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_${T}.log 2>&1; tail -1 $O/smoke_${T}.log
fi
if [[ " $PARTS " == *" bench "* ]]; then
  rm -f $O/${T}_bench_configs.jsonl $O/${T}_bench_stepham.jsonl $O/${T}_bench_wave.jsonl
  for sys in doublePendulum twoBody spring threeBodyPolar chain8 chain16 chain32; do
    timeout 500 python bench.py --system $sys --steps 20 --warmup 5 --cpu-seconds 5 2> $O/bench_${T}_${sys}.err | tail -1 >> $O/${T}_bench_configs.jsonl
    tail -1 $O/${T}_bench_configs.jsonl | head -c 140; echo
  done
  for sys in doublePendulum twoBody spring threeBodyPolar chain8 chain16; do
    timeout 300 python bench.py --integrator stepham --system $sys --steps 20 --warmup 3 2> $O/bench_${T}_stepham_${sys}.err | tail -1 >> $O/${T}_bench_stepham.jsonl
    tail -1 $O/${T}_bench_stepham.jsonl | head -c 170; echo
  done
  timeout 300 python bench.py --integrator stepham --system chain32 --batch 16384 --dt 0.02 --steps 10 --warmup 2 2>> $O/bench_${T}_stepham_chain32.err | tail -1 >> $O/${T}_bench_stepham.jsonl
  for sys in dense24 dense32 chain48 chain64; do
    timeout 300 python bench.py --system $sys --batch 16384 --rk4-per-step 20 --steps 10 --warmup 2 --cpu-seconds 3 2> $O/bench_${T}_${sys}.err | tail -1 >> $O/${T}_bench_wave.jsonl
    tail -1 $O/${T}_bench_wave.jsonl | head -c 140; echo
  done
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --force-dist --steps 20 --warmup 5 --no-cpu-baseline 2> $O/bench_${T}_dist.err | grep "^{" | tail -1 > $O/${T}_bench_force_dist.json
  head -c 160 $O/${T}_bench_force_dist.json; echo
fi
if [[ " $PARTS " == *" prof "* ]]; then
  for sys in doublePendulum twoBody spring threeBodyPolar chain8 chain16 chain32; do
    HAMK_PROF_PASSES="stats fetch write sq wait" timeout 400 bash scripts/profile.sh $T $sys > /dev/null 2>&1
  done
  for sys in dense32 chain64; do
    HAMK_PROF_PASSES="stats fetch write sq lds mfma wait" timeout 400 bash scripts/profile.sh $T $sys --batch 16384 --rk4-per-step 20 > /dev/null 2>&1
  done
  for sys in chain8 chain16; do
    HAMK_PROF_PASSES="stats sq fetch write lds wait" timeout 400 bash scripts/profile_stepham.sh $T $sys > /dev/null 2>&1
  done
  ls $O | grep prof_${T}
fi
