#!/bin/bash
# round 3, GPU call 5: backward kernels against the reference gradients; fused kernel after the scalar clean-up; tap-order probes
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c5
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 900 python -m pytest tests/test_grad_hip.py tests/test_abi.py -m gpu -q -x --timeout 600 -p no:cacheprovider -s > $OUT/pytest_grad.log 2>&1; echo "pytest grad rc=$?"; grep -E "passed|failed|worst deviation|Error|error" $OUT/pytest_grad.log | tail -12
timeout 400 python tools/bench_fused.py 100 0 5 18 19 15 100 > $OUT/bench_fused.log 2>&1; echo "bench_fused rc=$?"; grep -E "ABL|dev\(0\)" $OUT/bench_fused.log
