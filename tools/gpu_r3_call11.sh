#!/bin/bash
# round 3, GPU call 11: wave-task gather kernel against the per-float4 one
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c11
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 300 python tools/bench_gather.py 0 1 7 5 7 1 > $OUT/bench_gather.log 2>&1; echo "bench_gather rc=$?"; grep -E "cfg|Error|error" $OUT/bench_gather.log
