#!/bin/bash
# round 3, GPU call 15: gradient fixtures incl. the no-repeat case, the training entry point, a training step at the reference's shape
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c15
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 1200 python -m pytest tests/test_grad_hip.py tests/test_harness.py -m gpu -q --timeout 900 -p no:cacheprovider -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR|worst deviation|^E  " $OUT/pytest.log | tail -14
timeout 900 python experiment_scripts/train_realestate10k.py --experiment_name bench --views 2 --batch_size 12 --img_sidelength 256 --max_steps 8 --steps_til_summary 1 --logging_root $OUT/logs > $OUT/train256.log 2>&1; echo "train rc=$?"; grep -E "step|trained|Error" $OUT/train256.log | tail -10
rm -rf $OUT/logs
