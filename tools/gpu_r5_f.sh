#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r5f
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --maxfail=30 --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -30
timeout 600 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"
tail -1 $O/bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_without_partial_sums'], d['stage_ms'], d.get('pair_setup_ms'), d['eval_mode']['ms_per_step'], d['gather_stage']['frac'], d['rank_share']['projected_scaling_8'], d['first_round_ab'])"
timeout 600 python bench.py --config c3 --cpu-rays 0 > $O/bench_c3.log 2>&1; echo "bench c3 rc=$?"; tail -1 $O/bench_c3.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('pair_setup_ms'), d['eval_mode']['ms_per_step'])"
