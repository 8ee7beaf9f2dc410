#!/bin/bash
# round 3, GPU call 16: a training step at the reference's shape (12 scenes x 192 rays, 256x256, 64 samples), timed and traced
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c16
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 900 python experiment_scripts/train_realestate10k.py --experiment_name bench --views 2 --batch_size 12 --img_sidelength 256 --max_steps 12 --steps_til_summary 11 --logging_root $OUT/logs > $OUT/train256.log 2>&1; echo "train rc=$?"; grep -E "step|trained|Error" $OUT/train256.log | tail -5
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o train -- python $ROOT/experiment_scripts/train_realestate10k.py --experiment_name bench --views 2 --batch_size 12 --img_sidelength 256 --max_steps 4 --steps_til_summary 100 --logging_root $OUT/logs > $OUT/trace.log 2>&1; echo "trace rc=$?"
cd $ROOT
python - <<'PY'
import sqlite3, glob, re
hits = glob.glob('gpurun_out/r3c16/trace/**/*_results.db', recursive=True)
c = sqlite3.connect(hits[0]).cursor()
rows = list(c.execute("select name, count(*), avg(duration), sum(duration) from kernels group by name order by 4 desc"))
tot = sum(r[3] for r in rows)
print("| kernel | launches | avg us | total ms | % |\n|---|---|---|---|---|")
for n, k, a, t in rows[:28]:
    n = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", n)); n = re.sub(r"\(.*$", "", n)[:70]
    print(f"| `{n}` | {k} | {a/1e3:.1f} | {t/1e6:.2f} | {100*t/tot:.1f} |")
print(f"total kernel time {tot/1e6:.1f} ms over 4 steps (+ set-up)")
PY
rm -rf $OUT/logs; find $OUT -name "*.db" -size +3M -delete
