#!/bin/bash
# round 3, GPU call 7: phase stamps of the four-tap kernel
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c7
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 400 python tools/bench_fused.py 4 0 11 13 > $OUT/bench_fused.log 2>&1; echo "bench_fused rc=$?"; grep -v "amdgpu\|Warn" $OUT/bench_fused.log | tail -20
