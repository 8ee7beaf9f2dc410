#!/bin/bash
# round 3, GPU call 10: attend with non-temporal loads (stage times), re-run of the adjusted tests
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c10
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 400 python bench.py --cpu-rays 0 > $OUT/bench.log 2>&1; echo "bench rc=$?"; tail -1 $OUT/bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'], d['gather_stage']['frac'])"
timeout 900 python -m pytest tests/test_grad_hip.py tests/test_hip_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -k "gradient_step or device_made_poses" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR|^E  " $OUT/pytest.log | tail -10
