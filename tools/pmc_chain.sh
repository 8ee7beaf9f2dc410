#!/bin/bash
# PMC pass over the per-ray chain kernels (8192-ray forward calls of the default route, tools/bench_variants.py no_repeat is NOT used: default only)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_chain
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $OUT/sq -o p -- python $ROOT/tools/bench_variants.py no_sample > $OUT/sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace -d $OUT/lds -o p -- python $ROOT/tools/bench_variants.py no_sample > $OUT/lds.log 2>&1
cd $ROOT
python - <<'PY'
import sqlite3, glob
for d in sorted(glob.glob('gpurun_out/pmc_chain/*/p_results.db')):
    c = sqlite3.connect(d).cursor()
    print(d)
    for r in c.execute("select substr(kernel_name, 1, 44), counter_name, count(*), avg(value) from counters_collection where kernel_name like '%ray_tail%' or kernel_name like '%ray_mid%' or kernel_name like '%round2%' or kernel_name like '%attend%' group by 1, 2"):
        print("   %-46s %-28s n=%d avg=%.4g" % r)
    for r in c.execute("select substr(name,1,44), count(*), avg(duration) from kernels where name like '%ray_tail%' or name like '%ray_mid%' or name like '%round2%' or name like '%attend%' group by 1"):
        print("   %-46s n=%d avg %.1f us" % (r[0], r[1], r[2] / 1e3))
PY
find $OUT -name "*.db" -size +3M -delete
