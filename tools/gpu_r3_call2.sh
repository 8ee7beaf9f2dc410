#!/bin/bash
# round 3, GPU call 2: the GPU test suite on today's tree; gather-alone ablations with deeper tap prefetch; 2-D ray tiles; L1-miss latency PMC
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c2
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 --timeout 600 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest.log | tail -3
timeout 300 python tools/bench_fused.py 5 15 16 0 > $OUT/bench_fused.log 2>&1; echo "bench_fused rc=$?"; grep ABL $OUT/bench_fused.log
for t in 2x8 4x4 8x2 4x12 2x24; do
  CAR_BENCH_TILE=$t timeout 200 python tools/bench_fused.py 0 5 > $OUT/tile_$t.log 2>&1; echo "tile $t rc=$?"; grep ABL $OUT/tile_$t.log
done
cd /tmp
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --kernel-trace -d $OUT/lat0 -o p -- python $ROOT/tools/bench_fused.py 0 5 > $OUT/lat0.log 2>&1; echo "lat0 rc=$?"
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --kernel-trace -d $OUT/tcc0 -o p -- python $ROOT/tools/bench_fused.py 0 5 > $OUT/tcc0.log 2>&1; echo "tcc0 rc=$?"
cd $ROOT
python - <<'PY'
import sqlite3, glob
for d in sorted(glob.glob('gpurun_out/r3c2/**/*_results.db', recursive=True)):
    c = sqlite3.connect(d).cursor()
    print(d)
    try:
        for r in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%fused%' group by 1, 2"):
            print("   %-60s %-34s n=%d avg=%.6g" % (r[0][-60:], r[1], r[2], r[3]))
    except Exception as e:
        print("  ", e)
PY
find $OUT -name "*.db" -size +3M -delete
