#!/bin/bash
# round 3, GPU call 4: why the burst-issued gather (ABL 15) runs twice as fast as the slot-structured one (ABL 5): TCP / SQ counters of both
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c4
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE --kernel-trace -d $OUT/tcp -o p -- python $ROOT/tools/bench_fused.py 5 15 0 > $OUT/tcp.log 2>&1; echo "tcp rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $OUT/sq -o p -- python $ROOT/tools/bench_fused.py 5 15 0 > $OUT/sq.log 2>&1; echo "sq rc=$?"
cd $ROOT
python - <<'PY'
import sqlite3, glob
for d in sorted(glob.glob('gpurun_out/r3c4/**/*_results.db', recursive=True)):
    c = sqlite3.connect(d).cursor()
    print(d)
    for r in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%fused%' group by 1, 2"):
        print("   %-28s %-34s n=%d avg=%.6g" % (r[0][-46:-18], r[1], r[2], r[3]))
PY
find $OUT -name "*.db" -size +3M -delete
