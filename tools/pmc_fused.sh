#!/bin/bash
# PMC passes over the fused per-sample kernel alone (tools/bench_fused.py): SQ wave-state counters, TA/TCP and LDS activity.
# usage: tools/pmc_fused.sh [ablation variants...]   (default: 0)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_fused
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
VARS=${@:-0}
for abl in $VARS; do
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU --kernel-trace -d $OUT/sq$abl -o p -- python $ROOT/tools/bench_fused.py $abl > $OUT/sq$abl.log 2>&1
  timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --kernel-trace -d $OUT/tcp$abl -o p -- python $ROOT/tools/bench_fused.py $abl > $OUT/tcp$abl.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD TA_TA_BUSY_sum --kernel-trace -d $OUT/lds$abl -o p -- python $ROOT/tools/bench_fused.py $abl > $OUT/lds$abl.log 2>&1
done
cd $ROOT
python - <<'PY'
import sqlite3, glob
for d in sorted(glob.glob('gpurun_out/pmc_fused/*/p_results.db')):
    c = sqlite3.connect(d).cursor()
    print(d)
    for r in c.execute("select substr(kernel_name, 1, 40), counter_name, count(*), avg(value) from counters_collection where kernel_name like '%fused%' group by 1, 2"):
        print("   %-42s %-34s n=%d avg=%.4g" % r)
PY
find $OUT -name "*.db" -size +3M -delete
