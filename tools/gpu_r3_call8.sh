#!/bin/bash
# round 3, GPU call 8: key-layer halves after each source pass (no e_0 round trip inside the layer); gather-stage configurations
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c8
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 400 python tools/bench_fused.py 100 4 100 > $OUT/bench_fused.log 2>&1; echo "bench_fused rc=$?"; grep -v "amdgpu\|Warn" $OUT/bench_fused.log | tail -16
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "forward_matches or one_call or ragged or whole_frame or dynamic_range or zero" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest.log | tail -8
timeout 300 python tools/bench_gather.py > $OUT/bench_gather.log 2>&1; echo "bench_gather rc=$?"; grep cfg $OUT/bench_gather.log
