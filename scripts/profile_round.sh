#!/bin/bash
# Round profile on the GPU box: kernel trace + stats, then HBM/SQ counters in their own
# passes (never combined with a trace domain other than --kernel-trace).
# usage: bash scripts/profile_round.sh r01
set -o pipefail
TAG=${1:-r01}
OUT=/root/repo/gpurun_out/prof_$TAG
rm -rf $OUT   # NOTE: gpurun MERGES into the local gpurun_out/: delete the local copy before the call too
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
MAIN="python3 /root/repo/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra"
FULL="python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $MAIN > $OUT/bench_trace.log 2>&1; echo "trace rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $MAIN > $OUT/bench_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $MAIN > $OUT/bench_write.log 2>&1; echo "write rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $MAIN > $OUT/bench_sq.log 2>&1; echo "sq rc=$?"
# the extras (8 chains per GPU, residual kernel at C4 and HBM scale)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/x_trace -- $FULL > $OUT/bench_x_trace.log 2>&1; echo "x trace rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/x_pmc_fetch -- $FULL > $OUT/bench_x_fetch.log 2>&1; echo "x fetch rc=$?"
tail -1 $OUT/bench_trace.log | cut -c1-400
du -sh $OUT
