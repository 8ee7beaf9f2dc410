#!/bin/bash
# round 3, GPU call 1: TA microbenchmark, full-lattice timing probe (ABL 14) with PMC, baseline bench on this box
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c1
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 120 tools/_dev/ta_probe > $OUT/ta_probe.log 2>&1; echo "ta_probe rc=$?"; cat $OUT/ta_probe.log
timeout 300 python tools/bench_fused.py 0 14 5 0 > $OUT/bench_fused.log 2>&1; echo "bench_fused rc=$?"; grep ABL $OUT/bench_fused.log
cd /tmp
for abl in 0 14; do
  timeout 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr GRBM_GUI_ACTIVE TA_TA_BUSY_sum --kernel-trace -d $OUT/tcp$abl -o p -- python $ROOT/tools/bench_fused.py $abl > $OUT/tcp$abl.log 2>&1; echo "tcp$abl rc=$?"
  timeout 200 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/fetch$abl -o p -- python $ROOT/tools/bench_fused.py $abl > $OUT/fetch$abl.log 2>&1; echo "fetch$abl rc=$?"
done
cd $ROOT
python - <<'PY'
import sqlite3, glob
for d in sorted(glob.glob('gpurun_out/r3c1/**/*_results.db', recursive=True)):
    c = sqlite3.connect(d).cursor()
    print(d)
    try:
        for r in c.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%fused%' group by 1"):
            print("   %-34s n=%d avg=%.6g" % r)
    except Exception as e:
        print("  ", e)
PY
find $OUT -name "*.db" -size +3M -delete
timeout 400 python bench.py --cpu-rays 0 > $OUT/bench.log 2>&1; echo "bench rc=$?"; tail -1 $OUT/bench.log | cut -c1-600
