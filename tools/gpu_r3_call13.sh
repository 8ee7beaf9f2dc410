#!/bin/bash
# round 3, GPU call 13: scheduling variants of the tap loads inside a chunk
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c13
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 400 python tools/bench_fused.py 20 100 > $OUT/bench_fused.log 2>&1; echo "bench_fused rc=$?"; grep -v "amdgpu\|Warn" $OUT/bench_fused.log | tail -22
