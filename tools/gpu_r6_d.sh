#!/bin/bash
# round 6: (c) the step-major workgroup order on the whole frame (A/B/A/B on one box), (a) the 32x32x16 candidate against the product
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r6d
mkdir -p $O
for rep in 1 2; do for o in 0 2; do
  CAR_DEV_FLAGS="-DCAR_WG_ORDER=$o -DCAR_ABLATION_NONE" timeout 600 python tools/run_with_dev_lib.py bench.py --no-extras --cpu-rays 0 --steps 20 > $O/order${o}_$rep.log 2>&1
  tail -1 $O/order${o}_$rep.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('order $o rep $rep: ms', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['stage_ms'].items()})"
done; done
CAR_DEV_UNIT=car_fused_w32.hip CAR_DEV_FLAGS="-DCAR_ABLATION_NONE" timeout 900 python tools/bench_fused.py 100 400 100 400 > $O/w32.log 2>&1; echo "w32 rc=$?"; grep -E "ABL=|vs|rror" $O/w32.log
