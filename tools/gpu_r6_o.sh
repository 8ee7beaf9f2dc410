#!/bin/bash
# round 6 (second session): per-ray chains with the weight stream two chunks ahead; whole GPU suite; bench c2 with the rank share
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r6o
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 900 python bench.py --cpu-rays 0 > $O/bench_c2.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_c2.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step'],3), d['stage_ms'])
print('share', {k: (round(v,3) if isinstance(v,float) else v) for k,v in d['rank_share'].items() if k!='note'})
print('eval', d['eval_mode']['ms_per_step'], 'frac', d['roofline']['frac'])
"
