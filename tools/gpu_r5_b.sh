#!/bin/bash
# round 5, second GPU call: suite (with the out-of-bounds harness), bench, energy table, counter passes, harness self-check (last: it faults on purpose)
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r5b
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --maxfail=15 --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -20
timeout 600 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"
tail -1 $O/bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms'], d.get('pair_setup_ms'), d['rank_share'], d['power'])"
timeout 400 python tools/energy_probe.py 2.0 > $O/energy.log 2>&1; echo "energy rc=$?"; tail -16 $O/energy.log
timeout 900 bash tools/profile_bench.sh r5c2 > $O/profile.log 2>&1; echo "profile rc=$?"; tail -5 $O/profile.log
timeout 600 bash tools/pmc_fused.sh 101 > $O/pmc_fused.log 2>&1; echo "pmc rc=$?"; tail -40 $O/pmc_fused.log
timeout 600 bash tools/oob_selfcheck.sh > $O/oob_selfcheck.log 2>&1; echo "selfcheck rc=$?"; tail -15 $O/oob_selfcheck.log
