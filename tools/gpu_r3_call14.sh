#!/bin/bash
# round 3, GPU call 14: cache-policy bits on the tap loads
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c14
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
for f in "-DCAR_TAP_AUX=2" "-DCAR_TAP_AUX=1" "-DCAR_TAP_AUX=16" "-DCAR_TAP_AUX=3" ""; do
  CAR_DEV_FLAGS="$f" timeout 600 python tools/bench_fused.py 0 100 0 > $OUT/bench.log 2>&1; echo "flags '$f' rc=$?"; grep -E "^ABL|dev\(0\) vs product\(100\) e " $OUT/bench.log
done
