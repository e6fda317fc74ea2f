#!/bin/bash
# GPU box: rocprofv3 passes for one bench workload.  usage: scripts/profile.sh <tag> [system] [bench args...]
# Kernel trace + stats in one run; PMC counters each in their own run (never combined with sys/runtime
# traces).  Results land in gpurun_out/prof_<tag>_<system>/ ; scripts/summarize_profile.py turns them
# into profiles/<tag>_<system>_* and profiles/pmc_traffic_<system>.json.
set -u
export HAMK_TEST_OVERRIDES=1   # HAMK_SELFCHECK / HAMK_HIPRTC_FLAGS below are test overrides: read only when asked for
TAG=${1:-r02}
SYS=${2:-doublePendulum}
shift 2 2>/dev/null
EXTRA="$@"
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_$SYS
mkdir -p $OUT
export TMPDIR=/tmp
# the first-use self-check launches the same kernels on a few hundred trajectories: keep those tiny
# dispatches out of the per-kernel averages
export HAMK_SELFCHECK=0
cd /tmp
# HAMK_PROF_PASSES="stats fetch write sq lds mfma wait vmem ifetch": which passes to run (default: all)
want() { [ -z "${HAMK_PROF_PASSES:-}" ] || [[ " $HAMK_PROF_PASSES " == *" $1 "* ]]; }
B="python $R/bench.py --system $SYS --no-cpu-baseline $EXTRA"
want stats && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- $B --steps 20 --warmup 3 > $OUT/stats.log 2>&1
want fetch && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- $B --no-isa --steps 3 --warmup 1 > $OUT/pmc_fetch.log 2>&1
want write && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- $B --no-isa --steps 3 --warmup 1 > $OUT/pmc_write.log 2>&1
want sq && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o $TAG -- $B --no-isa --steps 3 --warmup 1 > $OUT/pmc_sq.log 2>&1
# LDS and matrix-core activity, each group in its own pass (a counter this build of rocprofv3 does not know costs only its group)
want lds && rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_lds -o $TAG -- $B --no-isa --steps 3 --warmup 1 > $OUT/pmc_lds.log 2>&1
want mfma && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_mfma -o $TAG -- $B --no-isa --steps 3 --warmup 1 > $OUT/pmc_mfma.log 2>&1
# where a wavefront's cycles go: SQ_WAVE_CYCLES = SQ_ACTIVE_INST_ANY + SQ_WAIT_ANY (parked at s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stalls)
want wait && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc_wait -o $TAG -- $B --no-isa --steps 3 --warmup 1 > $OUT/pmc_wait.log 2>&1
grep -h '"metric"' $OUT/stats.log | tail -1 > $OUT/bench_under_profiler.json
ls $OUT/*
