#!/bin/bash
# GPU box: rocprofv3 passes for the bench workload (config 2).  Kernel trace + stats in one run;
# PMC counters each in their own run (never combined with sys/runtime traces).  Results land in
# gpurun_out/prof_<tag>/ ; scripts/summarize_profile.py turns them into profiles/<tag>_*.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# the first-use self-check launches the same kernels on a few hundred trajectories: keep those tiny
# dispatches out of the per-kernel averages
export HAMK_SELFCHECK=0
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- python $R/bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o $TAG -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc_sq.log 2>&1
grep -h '"metric"' $OUT/stats.log | tail -1 > $OUT/bench_under_profiler.json
ls $OUT/*
