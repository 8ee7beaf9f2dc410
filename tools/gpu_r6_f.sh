#!/bin/bash
# round 6: eval-loop prefetch (item 6) — parity test, bench line with the eval A/B at c2 and c3, the eval script itself
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r6f
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_harness.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "prefetched or entry_points or bench_prints or nview3 or exchange" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for c in c2 c3; do timeout 900 python bench.py --config $c --cpu-rays 0 > $O/bench_$c.log 2>&1; echo "bench $c rc=$?"; tail -1 $O/bench_$c.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$c', round(d['ms_per_step'],2), 'eval', d['eval_mode']['ms_per_step'], d['eval_mode']['ms_per_step_without_prefetch'], 'setup', d['pair_setup_ms'])"; done
timeout 600 python experiment_scripts/eval_realestate10k.py --experiment_name t --views 2 --synthetic --batch_size 4 > $O/eval.log 2>&1; echo "eval rc=$?"; grep -E "item|mean" $O/eval.log
