#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of one bench step, then separate PMC passes for the
# HBM traffic counters (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC has 4 slots, they cost 3 + 2).
# Usage: tools/profile_bench.sh <tag>     -> gpurun_out/prof_<tag>/{trace,fetch,write}/..., summaries in gpurun_out/
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $ROOT/bench.py --steps 1 --warmup 1 --cpu-rays 0"
PMC_EXTRA=${PMC_EXTRA:---no-extras}     # the PMC passes count bytes per frame: no gather-stage / per-rank extras
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $BENCH ${TRACE_EXTRA:-} > $OUT/trace.log 2>&1
echo "trace rc=$?"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o bench -- $BENCH $PMC_EXTRA > $OUT/fetch.log 2>&1
echo "fetch rc=$?"
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o bench -- $BENCH $PMC_EXTRA > $OUT/write.log 2>&1
echo "write rc=$?"
cd $ROOT
python tools/summarize_prof.py $OUT --json $ROOT/gpurun_out/traffic_${TAG}.json > $ROOT/gpurun_out/prof_${TAG}_summary.md 2>&1
echo "summary rc=$?"
# keep the merged payload small: drop the raw per-dispatch traces, keep stats + summaries
find $OUT -name "*kernel_trace*" -size +2M -delete
du -sh $OUT
