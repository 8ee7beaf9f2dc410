#!/bin/bash
# PMC passes over the split-fp16 linear kernel (tools/bench_linear.py): wave-state and LDS counters per kernel instance
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_linear
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $OUT/sq -o p -- python $ROOT/tools/bench_linear.py > $OUT/sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace -d $OUT/lds -o p -- python $ROOT/tools/bench_linear.py > $OUT/lds.log 2>&1
cd $ROOT
python - <<'PY'
import sqlite3, glob
for d in sorted(glob.glob('gpurun_out/pmc_linear/*/p_results.db')):
    c = sqlite3.connect(d).cursor()
    print(d)
    for r in c.execute("select substr(kernel_name, 1, 50), grid_size_x, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%linear16%' group by 1, 2, 3"):
        print("   %-52s %9d %-28s n=%d avg=%.4g" % r)
PY
find $OUT -name "*.db" -size +3M -delete
