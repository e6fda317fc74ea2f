#!/bin/bash
# GPU box: rocprofv3 passes for the reference's own stepper (hamk_rkf45_k, bench.py --integrator stepham).
# usage: scripts/profile_stepham.sh <tag> <system> [bench args...]; results in gpurun_out/prof_<tag>_<system>_stepham/
set -u
export HAMK_TEST_OVERRIDES=1   # HAMK_SELFCHECK / HAMK_HIPRTC_FLAGS below are test overrides: read only when asked for
TAG=${1:-r03}
SYS=${2:-doublePendulum}
shift 2 2>/dev/null
EXTRA="$@"
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_${SYS}_stepham
mkdir -p $OUT
export TMPDIR=/tmp
export HAMK_SELFCHECK=0
cd /tmp
# HAMK_PROF_PASSES="stats fetch write sq lds mfma wait vmem ifetch": which passes to run (default: all)
want() { [ -z "${HAMK_PROF_PASSES:-}" ] || [[ " $HAMK_PROF_PASSES " == *" $1 "* ]]; }
B="python $R/bench.py --system $SYS --integrator stepham --no-cpu-baseline $EXTRA"
want stats && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- $B --steps 20 --warmup 3 > $OUT/stats.log 2>&1
want sq && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o $TAG -- $B --no-isa --steps 3 --warmup 1 > $OUT/pmc_sq.log 2>&1
want fetch && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- $B --no-isa --steps 3 --warmup 1 > $OUT/pmc_fetch.log 2>&1
want write && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- $B --no-isa --steps 3 --warmup 1 > $OUT/pmc_write.log 2>&1
# where the wavefronts wait: LDS and vector-memory activity, each group in its own pass (a counter this build of rocprofv3 does not know costs only its group)
want lds && rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_lds -o $TAG -- $B --no-isa --steps 3 --warmup 1 > $OUT/pmc_lds.log 2>&1
want vmem && rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_vmem -o $TAG -- $B --no-isa --steps 3 --warmup 1 > $OUT/pmc_vmem.log 2>&1
want ifetch && rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_ifetch -o $TAG -- $B --no-isa --steps 3 --warmup 1 > $OUT/pmc_ifetch.log 2>&1
grep -h '"metric"' $OUT/stats.log | tail -1 > $OUT/bench_under_profiler.json
ls $OUT
