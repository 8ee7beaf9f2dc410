#!/bin/bash
# round 6 (second session): kernel trace of the three-view route (8192-ray calls)
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r6q
mkdir -p $O
timeout 600 python tools/bench_variants.py nview3 nview1 no_latent_concat > $O/variants.log 2>&1; tail -4 $O/variants.log
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o nv3 -- python $GRAFT_REPO_ROOT/tools/bench_variants.py nview3 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1; cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3, re, glob
for f in glob.glob('gpurun_out/r6q/prof/**/*_results.db', recursive=True):
    c = sqlite3.connect(f).cursor()
    rows = list(c.execute("select name, count(*), avg(duration), sum(duration) from kernels group by name order by 4 desc"))
    print("total ms", sum(r[3] for r in rows) / 1e6, "calls of the forward: 18 (2 x 6 warm-up + 6 timed)")
    for name, n, avg, s in rows[:30]:
        print(f"{s/1e6/18:8.3f} ms/call {n/18:6.1f} x {avg/1e3:8.1f} us  {re.sub(r'.anonymous namespace.::', '', name)[:100]}")
PY
