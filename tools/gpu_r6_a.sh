#!/bin/bash
# round 6, first GPU call: the second round without stored query rows (car_round2_logits_from_g) — parity, out-of-bounds harness, A/B bench line
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r6a
mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_fused_pack.py tests/test_oob_guard.py tests/test_abi.py -m gpu -q -x --timeout 900 -p no:cacheprovider \
  -k "bilinear or packers or oob or round2 or one_call_route or two_phase or first_round or forward_matches or without_second_round or full_size or dynamic_range or all_zero" > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR|Error" $O/pytest.log | tail -15
timeout 900 python bench.py --cpu-rays 0 > $O/bench_c2.log 2>&1; echo "bench rc=$?"
tail -1 $O/bench_c2.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'])
print('stage', d['stage_ms'])
print('frac_executed', d['roofline']['frac_executed'])
print('ab1', d['first_round_ab']['rows_of_e']['ms_per_step'])
print('eval', d['eval_mode']['ms_per_step'], 'setup', d['pair_setup_ms'], 'ws', d['workspace_bytes'], 'share', d['rank_share'])
"
