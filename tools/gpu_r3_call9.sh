#!/bin/bash
# round 3, GPU call 9: non-temporal output stores in the fused kernel; the whole GPU suite (with the new tests)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c9
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
for f in "-DCAR_NT=1" "-DCAR_NT=2" ""; do
  CAR_DEV_FLAGS="$f" timeout 600 python tools/bench_fused.py 0 100 0 > $OUT/bench_nt.log 2>&1; echo "flags '$f' rc=$?"; grep -E "^ABL" $OUT/bench_nt.log
done
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 --timeout 900 -p no:cacheprovider -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR|device poses vs reference fixture|worst deviation" $OUT/pytest.log | tail -40
