#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r5train
mkdir -p $O
python tools/bench_wgrad.py 2>&1 | grep "M=" | tee $O/wgrad.log
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/trace -o t -- python $OLDPWD/tools/train_step_probe.py > $OLDPWD/$O/train_prof.log 2>&1); echo "trace rc=$?"; tail -3 $O/train_prof.log
python - <<'PY'
import sqlite3, glob, re
for d in glob.glob('gpurun_out/r5train/trace/**/*_results.db', recursive=True):
    c = sqlite3.connect(d).cursor()
    rows = list(c.execute("select name, grid_x, workgroup_x, count(*), avg(duration), sum(duration) from kernels group by name, grid_x order by 6 desc"))
    tot = sum(r[5] for r in rows)
    print("| kernel | grid x workgroup | launches | avg us | total ms | % |\n|---|---|---|---|---|---|")
    for n, g, w, k, a, s in rows[:28]:
        n = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", n)); n = re.sub(r"\(.*$", "", n)
        print(f"| `{n[:110]}` | {g} x {w} | {k} | {a/1e3:.1f} | {s/1e6:.2f} | {100*s/tot:.1f} |")
    print(f"\nTotal kernel time {tot/1e6:.1f} ms.")
PY
find $O/trace -name "*.db" -size +3M -delete
timeout 600 python experiment_scripts/train_realestate10k.py --help > /dev/null 2>&1; echo help rc=$?
