set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r6final3
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --maxfail=30 --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -10
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; grep -E "smoke|hipcc" $O/smoke.log
timeout 900 python bench.py > $O/bench_c2.log 2>&1; echo "bench rc=$?"
tail -1 $O/bench_c2.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms'], d.get('pair_setup_ms'), d['eval_mode']['ms_per_step'], d['rank_share']['projected_scaling_8'], d['rank_share']['ms_per_step_8'])"
for c in c3 c4 c5; do timeout 600 python bench.py --config $c --cpu-rays 0 > $O/bench_$c.log 2>&1; echo "bench $c rc=$?"; tail -1 $O/bench_$c.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms'])"; done
timeout 300 python tools/train_step_probe.py 12 > $O/train.log 2>&1; tail -3 $O/train.log
