#!/bin/bash
# round 6 (second session): ReLU mask in the data-gradient kernel's store, one zero fill for the parameter gradients
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r6k
mkdir -p $O
timeout 1500 python -m pytest tests/test_grad_hip.py tests/test_hip_parity.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "grad or linear or train or wgrad" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 600 python tools/train_step_probe.py 12 > $O/train.log 2>&1; echo "train rc=$?"; tail -4 $O/train.log
