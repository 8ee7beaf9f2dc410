#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r5e
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --maxfail=30 --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -30
timeout 600 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"
tail -1 $O/bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms'], d.get('pair_setup_ms'), d['eval_mode']['ms_per_step'], d['gather_stage']['frac'], d['rank_share']['projected_scaling_8'])"
timeout 600 python bench.py --config c3 --cpu-rays 0 > $O/bench_c3.log 2>&1; echo "bench c3 rc=$?"; tail -1 $O/bench_c3.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms'], d.get('pair_setup_ms'), d['eval_mode']['ms_per_step'], d['gather_stage']['frac'])"
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/setup -o v -- python $OLDPWD/bench.py --steps 2 --warmup 1 --cpu-rays 0 > $OLDPWD/$O/setup.log 2>&1); echo "trace rc=$?"
python - <<'PY'
import sqlite3, glob
for d in glob.glob('gpurun_out/r5e/setup/**/*_results.db', recursive=True):
    c = sqlite3.connect(d).cursor()
    for n, g, k, a in c.execute("select name, grid_x, count(*), avg(duration) from kernels where name like '%merge%' or name like '%linear16%' or name like '%linear_kernel%' group by name, grid_x"):
        print(f"{n[:80]:80s} grid {g:10d} n={k:4d} avg {a/1e3:9.1f} us")
PY
find $O/setup -name "*.db" -size +3M -delete
