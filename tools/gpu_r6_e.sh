#!/bin/bash
# round 6: energy / clock rows of the closing experiments — (a) 32x32x16 candidate, (c) ray-major order of rounds 2-5 against the product (step-major)
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r6e
mkdir -p $O
CAR_ENERGY_ROWS=closing CAR_DEV_UNIT=car_fused_w32.hip CAR_DEV_FLAGS="-DCAR_WG_ORDER=0 -DCAR_ABLATION_NONE" timeout 900 python tools/energy_probe.py 3 > $O/energy.log 2>&1; echo "energy rc=$?"
grep "^|" $O/energy.log
CAR_DEV_UNIT=car_fused_w32.hip CAR_DEV_FLAGS="-DCAR_ABLATION_NONE" timeout 900 python tools/bench_fused.py 100 400 100 400 > $O/w32.log 2>&1; echo "w32 rc=$?"; grep -E "ABL=|vs|rror" $O/w32.log
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "fused_rows or one_call_route or first_round or full_size or c3_twelve or chunks" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python bench.py --cpu-rays 0 > $O/bench_c2.log 2>&1; echo "bench rc=$?"
tail -1 $O/bench_c2.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['frac_executed'])
print('stage', d['stage_ms'])
print('eval', d['eval_mode']['ms_per_step'], 'setup', d['pair_setup_ms'], 'share', {k: round(v, 3) for k, v in d['rank_share'].items() if isinstance(v, float)})
"
for c in c3 c4 c5; do timeout 600 python bench.py --config $c --cpu-rays 0 --no-extras > $O/bench_$c.log 2>&1; tail -1 $O/bench_$c.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$c', round(d['ms_per_step'],2), round(d['roofline']['frac'],4), {k: round(v,2) for k,v in d['stage_ms'].items()})"; done
