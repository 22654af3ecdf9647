#!/bin/bash
# Round profile on the GPU box: kernel trace + stats, then HBM / SQ / MFMA counters in their own
# passes (a --pmc pass is never combined with a trace domain other than --kernel-trace).
# usage: bash scripts/profile_round.sh r02      (then: python3 scripts/summarize_profile.py r02)
set -o pipefail
TAG=${1:-r03}
OUT=/root/repo/gpurun_out/prof_$TAG
rm -rf $OUT   # NOTE: gpurun MERGES into the local gpurun_out/: delete the local copy before the call too
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the headline step exactly as bench.py runs it, and the other kernels at the BASELINE sizes
MAIN="python3 /root/repo/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra"
WORK="python3 /root/repo/scripts/profile_workload.py"
pass() {  # name, rocprofv3 options..., then the workload after --
    local name=$1; shift
    rocprofv3 "$@" > $OUT/$name.log 2>&1; echo "$name rc=$?"
}
pass trace       --kernel-trace --stats --output-format csv -d $OUT/trace -- $MAIN
pass pmc_fetch   --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $MAIN
pass pmc_write   --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $MAIN
pass pmc_sq      --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $MAIN
pass x_trace     --kernel-trace --stats --output-format csv -d $OUT/x_trace -- $WORK
pass x_pmc_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/x_pmc_fetch -- $WORK
pass x_pmc_write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/x_pmc_write -- $WORK
pass x_pmc_mfma  --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/x_pmc_mfma -- $WORK
tail -1 $OUT/trace.log | cut -c1-300
tail -8 $OUT/x_trace.log
# keep what gets merged back small: the per-dispatch traces of the long runs are not needed
find $OUT -name "*_agent_info.csv" -delete
du -sh $OUT
