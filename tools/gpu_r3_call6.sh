#!/bin/bash
# round 3, GPU call 6: full three-level lattice (4 taps per sample and source): GPU test suite, fused kernel timing, bench
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c6
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 400 python tools/bench_fused.py 100 1 2 5 100 > $OUT/bench_fused.log 2>&1; echo "bench_fused rc=$?"; grep -E "ABL|Error|error" $OUT/bench_fused.log | head
timeout 1200 python -m pytest tests -m gpu -q --maxfail=15 --timeout 600 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest.log | tail -20
timeout 400 python bench.py --cpu-rays 0 > $OUT/bench.log 2>&1; echo "bench rc=$?"; tail -1 $OUT/bench.log | cut -c1-1500
