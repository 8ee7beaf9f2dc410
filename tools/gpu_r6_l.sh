#!/bin/bash
# round 6 (second session): the binned scatter (grid_sample backward without float atomics)
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r6l
mkdir -p $O
timeout 1500 python -m pytest tests/test_grad_hip.py -m gpu -q -x --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
timeout 600 python tools/train_step_probe.py 12 > $O/train.log 2>&1; echo "train rc=$?"; tail -4 $O/train.log
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o train -- python $GRAFT_REPO_ROOT/tools/train_step_probe.py 10 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1; cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3, re, glob
for f in glob.glob('gpurun_out/r6l/prof/**/*_results.db', recursive=True):
    c = sqlite3.connect(f).cursor()
    rows = list(c.execute("select name, count(*), avg(duration), sum(duration) from kernels group by name order by 4 desc"))
    print("total ms", sum(r[3] for r in rows) / 1e6)
    for name, n, avg, s in rows[:24]:
        print(f"{s/1e6:8.2f} ms {n:6d} x {avg/1e3:8.1f} us  {re.sub(r'.anonymous namespace.::', '', name)[:90]}")
PY
