#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r5d
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --maxfail=30 --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -30
timeout 600 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"
tail -1 $O/bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms'], d.get('pair_setup_ms'), d['eval_mode']['ms_per_step'], d['gather_stage']['frac'], d['rank_share']['projected_scaling_8'])"
timeout 600 python tools/bench_variants.py > $O/variants.log 2>&1; echo "variants rc=$?"; cat $O/variants.log | grep -v amdgpu.ids
timeout 600 python bench.py --config c3 --cpu-rays 0 > $O/bench_c3.log 2>&1; echo "bench c3 rc=$?"; tail -1 $O/bench_c3.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms'], d.get('pair_setup_ms'), d['eval_mode']['ms_per_step'], d['gather_stage']['frac'])"
