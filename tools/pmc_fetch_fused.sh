#!/bin/bash
# Round 6, experiment (c): the fused kernel's memory-side reads (FETCH_SIZE) and time under other workgroup orders.
# usage: tools/pmc_fetch_fused.sh <tag> [bench_fused variants...]   (CAR_DEV_FLAGS / CAR_BENCH_TILE from the environment)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/fetch_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
VARS=${@:-0}
cd $ROOT
python tools/bench_fused.py $VARS > $OUT/time.log 2>&1; echo "[$TAG] time rc=$?"; grep "ABL=\|equal" $OUT/time.log
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o p -- python $ROOT/tools/bench_fused.py $VARS > $OUT/fetch.log 2>&1; echo "[$TAG] fetch rc=$?"
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/l2 -o p -- python $ROOT/tools/bench_fused.py $VARS > $OUT/l2.log 2>&1; echo "[$TAG] l2 rc=$?"
cd $ROOT
python - "$OUT" <<'PY'
import sqlite3, glob, sys
for d in sorted(glob.glob(sys.argv[1] + '/*/**/p_results.db', recursive=True)):
    c = sqlite3.connect(d).cursor()
    for r in c.execute("select substr(kernel_name, 1, 40), counter_name, count(*), avg(value) from counters_collection where kernel_name like '%fused_kernel%' group by 1, 2"):
        print("   %-42s %-16s n=%d avg=%.5g" % r)
PY
find $OUT -name "*.db" -size +3M -delete
