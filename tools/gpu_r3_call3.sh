#!/bin/bash
# round 3, GPU call 3: 8-wave ring-scheduled fused kernel (dev build) against the product's 12-wave kernel
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c3
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
export CAR_DEV_FLAGS="-DCAR_FUSED_WAVES=8"
cd $ROOT
timeout 400 python tools/bench_fused.py 100 0 1 2 5 100 0 > $OUT/bench_fused.log 2>&1; echo "bench_fused rc=$?"; grep -E "ABL|dev\(0\)" $OUT/bench_fused.log; tail -5 $OUT/bench_fused.log | grep -v ABL
