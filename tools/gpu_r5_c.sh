#!/bin/bash
# round 5, third GPU call: suite with the NaN-margin harness, its self-check, kernel trace of the n_view=3 route, the other configs' lines
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r5c
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --maxfail=30 --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -30
timeout 600 bash tools/oob_selfcheck.sh > $O/oob_selfcheck.log 2>&1; echo "selfcheck rc=$?"; cat $O/oob_selfcheck.log | tail -20
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/nview3 -o v -- python $OLDPWD/tools/bench_variants.py nview3 > $OLDPWD/$O/nview3.log 2>&1); echo "nview3 rc=$?"; tail -3 $O/nview3.log
python - <<'PY'
import sqlite3, glob
for d in glob.glob('gpurun_out/r5c/nview3/**/*_results.db', recursive=True):
    c = sqlite3.connect(d).cursor()
    rows = list(c.execute("select name, grid_x, count(*), avg(duration), sum(duration) from kernels group by name, grid_x order by 5 desc"))
    tot = sum(r[4] for r in rows)
    for n, g, k, a, s in rows[:25]:
        print(f"{n[:90]:90s} grid {g:10d} n={k:4d} avg {a/1e3:9.1f} us  total {s/1e6:8.2f} ms {100*s/tot:5.1f}%")
PY
find $O/nview3 -name "*.db" -size +3M -delete
for c in c3 c4 c5; do timeout 600 python bench.py --config $c --cpu-rays 0 > $O/bench_$c.log 2>&1; echo "bench $c rc=$?"; tail -1 $O/bench_$c.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms'], d.get('pair_setup_ms'), d['eval_mode']['ms_per_step'], d['gather_stage']['frac'])"; done
timeout 300 python tools/train_step_probe.py > $O/train.log 2>&1; echo "train rc=$?"; tail -5 $O/train.log
