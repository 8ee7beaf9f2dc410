#!/bin/bash
# round 3, GPU call 17: double-buffered weight-gradient kernel
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c17
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 600 python -m pytest tests/test_grad_hip.py -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^E  " $OUT/pytest.log | tail -5
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o train -- python $ROOT/experiment_scripts/train_realestate10k.py --experiment_name bench --views 2 --batch_size 12 --img_sidelength 256 --max_steps 4 --steps_til_summary 100 --logging_root $OUT/logs > $OUT/trace.log 2>&1; echo "trace rc=$?"
cd $ROOT
python - <<'PY'
import sqlite3, glob, re
hits = glob.glob('gpurun_out/r3c17/trace/**/*_results.db', recursive=True)
c = sqlite3.connect(hits[0]).cursor()
rows = list(c.execute("select name, count(*), avg(duration), sum(duration) from kernels group by name order by 4 desc"))
tot = sum(r[3] for r in rows)
for n, k, a, t in rows[:6]:
    n = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", n)); n = re.sub(r"\(.*$", "", n)[:50]
    print(f"| `{n}` | {k} | {a/1e3:.1f} | {t/1e6:.2f} | {100*t/tot:.1f} |")
print(f"total kernel time {tot/1e6:.1f} ms over 4 steps")
for r in c.execute("select grid_x, grid_y, grid_z, count(*), avg(duration) from kernels where name like '%wgrad%' group by 1,2,3 order by 5 desc"):
    print("   wgrad grid", r[0], r[1], r[2], "launches", r[3], "avg us %.1f" % (r[4]/1e3))
PY
rm -rf $OUT/logs; find $OUT -name "*.db" -size +3M -delete
