#!/bin/bash
# GPU box, round 6, the final-tree evidence pass -- ONE box:
#   part "bench":  one RK4 line per BASELINE config WITH cpu_baseline and parity, the reference's own stepper on all seven, the dense maps
#                  (four-lane kernels) and the wave-cooperative kernels, the RCCL path on one rank, the default bench line
#   part "prof":   rocprofv3 stats + PMC (HBM traffic -> profiles/pmc_traffic_*.json, SQ counters, wait split) for the headline kernel, the
#                  C3-C5 kernels, dense32 on the four-lane kernels, chain64 on the wave kernels, the stepper at n = 8, 16
#   part "tests":  the whole GPU suite + smoke()
# usage: scripts/gpu_r06_final.sh [bench] [prof] [tests]   (default: bench prof).  Everything is pre-compiled (scripts/warm_cache.py).
set -u
export HAMK_TEST_OVERRIDES=1
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export HAMK_CACHE_DIR=$PWD/.hamk_cache
O=gpurun_out
mkdir -p $O
PARTS="${@:-bench prof}"
T=r06
if [[ " $PARTS " == *" bench "* ]]; then
  rm -f $O/${T}_bench_configs.jsonl $O/${T}_bench_stepham.jsonl $O/${T}_bench_wave.jsonl $O/${T}_bench_dense.jsonl
  timeout 400 python bench.py 2> $O/bench_${T}_default.err | tail -n 1 > $O/${T}_bench_line.json
  for sys in doublePendulum twoBody spring threeBodyPolar chain8 chain16 chain32; do
    timeout 500 python bench.py --system $sys --steps 20 --warmup 5 --cpu-seconds 5 2> $O/bench_${T}_${sys}.err | tail -n 1 >> $O/${T}_bench_configs.jsonl
    tail -n 1 $O/${T}_bench_configs.jsonl | head -c 140; echo
  done
  for sys in doublePendulum twoBody spring threeBodyPolar chain8 chain16; do
    timeout 300 python bench.py --integrator stepham --system $sys --steps 20 --warmup 3 2> $O/bench_${T}_stepham_${sys}.err | tail -n 1 >> $O/${T}_bench_stepham.jsonl
  done
  timeout 300 python bench.py --integrator stepham --system chain32 --batch 16384 --dt 0.02 --steps 10 --warmup 2 2>> $O/bench_${T}_stepham_chain32.err | tail -n 1 >> $O/${T}_bench_stepham.jsonl
  for sys in dense18 dense24 dense32 denseD24 denseD32; do
    NOISA=""; [[ $sys == denseD* ]] && NOISA="--no-isa"      # (their instruction-count probe builds are not pre-compiled: minutes of hiprtc)
    timeout 300 python bench.py --system $sys --batch 16384 --rk4-per-step 20 --steps 10 --warmup 2 --cpu-seconds 3 $NOISA 2> $O/bench_${T}_${sys}.err | tail -n 1 >> $O/${T}_bench_dense.jsonl
  done
  for sys in chain48 chain64; do
    timeout 300 python bench.py --system $sys --batch 16384 --rk4-per-step 20 --steps 10 --warmup 2 --cpu-seconds 3 2> $O/bench_${T}_${sys}.err | tail -n 1 >> $O/${T}_bench_wave.jsonl
  done
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --force-dist --steps 20 --warmup 5 --no-cpu-baseline 2> $O/bench_${T}_dist.err | grep "^{" | tail -n 1 > $O/${T}_bench_force_dist.json
  timeout 300 python bench.py --gpus 2 --dist-backend gloo --steps 5 --warmup 2 --no-cpu-baseline 2> $O/bench_${T}_two_ranks.err | grep "^{" | tail -n 1 > $O/${T}_bench_two_ranks_one_gpu.json
  python - <<PY
import json, glob
for f in ("$O/${T}_bench_line.json", "$O/${T}_bench_configs.jsonl", "$O/${T}_bench_stepham.jsonl", "$O/${T}_bench_dense.jsonl", "$O/${T}_bench_wave.jsonl", "$O/${T}_bench_force_dist.json", "$O/${T}_bench_two_ranks_one_gpu.json"):
    for l in open(f):
        if l.startswith("{"):
            r = json.loads(l); rf = r.get("roofline", {}); fp = rf.get("fp64", {})
            print(f.split("/")[-1][:22], r["config"]["workload"][:30], r["config"].get("kernel_path", "")[:12], "n_gpus", r["n_gpus"], "%.4g" % r["value"], "8d", rf.get("frac") and round(rf["frac"], 3), "valu", fp.get("valu_insts_per_wave_step"),
                  "cpu", r.get("cpu_baseline", {}).get("value") and "%.3g" % r["cpu_baseline"]["value"], "par1", r.get("parity", {}).get("max_abs_dphase_1_step"), r.get("parity", {}).get("identical_substep_counts_frac"))
PY
fi
if [[ " $PARTS " == *" prof "* ]]; then
  for sys in doublePendulum twoBody spring threeBodyPolar chain8 chain16 chain32; do
    HAMK_PROF_PASSES="stats fetch write sq wait" timeout 400 bash scripts/profile.sh $T $sys > /dev/null 2>&1
  done
  HAMK_PROF_PASSES="stats fetch write sq wait" timeout 400 bash scripts/profile.sh $T dense32 --batch 16384 --rk4-per-step 20 > /dev/null 2>&1
  HAMK_PROF_PASSES="stats fetch write sq lds mfma wait" timeout 400 bash scripts/profile.sh $T chain64 --batch 16384 --rk4-per-step 20 > /dev/null 2>&1
  for sys in chain8 chain16; do
    HAMK_PROF_PASSES="stats sq fetch write lds wait" timeout 400 bash scripts/profile_stepham.sh $T $sys > /dev/null 2>&1
  done
  ls $O | grep prof_${T}
fi
if [[ " $PARTS " == *" tests "* ]]; then
  export HAMK_TEST_RECORD=$PWD/$O/${T}_gpu_test_record.jsonl
  rm -f $HAMK_TEST_RECORD
  timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/gputest_${T}.log 2>&1; echo "pytest rc=$?" >> $O/gputest_${T}.log
  tail -n 14 $O/gputest_${T}.log
  unset HAMK_TEST_RECORD
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_${T}.log 2>&1; tail -n 1 $O/smoke_${T}.log
fi
