#!/bin/bash
# round 6 (second session): column groups of the split-fp16 linear kernel on one XCD
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r6m
mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_grad_hip.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "linear or grad or lattice or encode or key_query or one_call or stage" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 600 python tools/train_step_probe.py 12 > $O/train.log 2>&1; echo "train rc=$?"; tail -4 $O/train.log
timeout 600 python tools/bench_linear.py > $O/bench_linear.log 2>&1; echo "bench_linear rc=$?"; tail -12 $O/bench_linear.log
timeout 900 python bench.py --no-extras --cpu-rays 0 > $O/bench_c2.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_c2.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['stage_ms'], d.get('pair_setup_ms'))"
