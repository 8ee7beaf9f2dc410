#!/bin/bash
# round 6, the closing build's records: suite, smoke, bench lines of every config, kernel trace + PMC passes, variants, whole-frame parity
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r6final
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --maxfail=30 --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -30
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; grep -E "smoke|hipcc" $O/smoke.log
timeout 900 python bench.py > $O/bench_c2.log 2>&1; echo "bench rc=$?"
tail -1 $O/bench_c2.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_executed'], d['stage_ms'], d.get('pair_setup_ms'), d['eval_mode']['ms_per_step'], d['gather_stage']['frac'], d['rank_share'], d['power'])"
for c in c3 c4 c5; do timeout 600 python bench.py --config $c --cpu-rays 0 > $O/bench_$c.log 2>&1; echo "bench $c rc=$?"; tail -1 $O/bench_$c.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_executed'], d['stage_ms'], d.get('pair_setup_ms'), d['eval_mode']['ms_per_step'], d['gather_stage']['frac'], d['lattice_bytes'], d['workspace_bytes'])"; done
timeout 900 bash tools/profile_bench.sh r6final > $O/profile.log 2>&1; echo "profile rc=$?"; tail -5 $O/profile.log
timeout 600 bash tools/pmc_fused.sh 101 > $O/pmc_fused.log 2>&1; echo "pmc rc=$?"; grep -A30 "sq101\|tcp101\|lds101" $O/pmc_fused.log | grep -E "p_results|fused" | head -40
timeout 600 python tools/bench_variants.py > $O/variants.log 2>&1; echo "variants rc=$?"; grep -v amdgpu.ids $O/variants.log
timeout 300 python tools/train_step_probe.py > $O/train.log 2>&1; echo "train rc=$?"; tail -3 $O/train.log
timeout 300 python tools/bench_merge.py > $O/merge.log 2>&1; grep merge_lattice $O/merge.log
timeout 900 python tools/validate_frame.py 0.5 0.1 > $O/whole_frame_parity.md 2> $O/validate.err; echo "validate rc=$?"; cat $O/whole_frame_parity.md
