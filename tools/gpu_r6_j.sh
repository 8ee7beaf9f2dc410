#!/bin/bash
# round 6 (second session): the rebuilt wgrad16 kernel — gradient tests, the out-of-bounds cases, stand-alone timing, the training step and its kernel trace
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r6j
mkdir -p $O
timeout 1500 python -m pytest tests/test_grad_hip.py tests/test_oob_guard.py -m gpu -q -x --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 300 python tools/bench_wgrad.py > $O/bench_wgrad.log 2>&1; echo "bench_wgrad rc=$?"; cat $O/bench_wgrad.log
timeout 600 python tools/train_step_probe.py 12 > $O/train.log 2>&1; echo "train rc=$?"; tail -4 $O/train.log
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o train -- python $GRAFT_REPO_ROOT/tools/train_step_probe.py 10 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1; cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r6j/prof/**/*kernel_stats.csv', recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    for r in rows[:40]:
        print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {int(r['Calls']):6d} calls {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:110]}")
PY
find $O/prof -name "*kernel_trace*" -size +2M -delete
