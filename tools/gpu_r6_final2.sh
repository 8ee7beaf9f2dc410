#!/bin/bash
# round 6, second session: the closing build's records — suite, smoke, bench lines of every config, kernel trace + traffic passes, variants,
# the training step with its kernel trace, the weight-gradient kernel alone
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r6final2
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --maxfail=30 --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -30
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; grep -E "smoke|hipcc" $O/smoke.log
timeout 900 python bench.py > $O/bench_c2.log 2>&1; echo "bench rc=$?"
tail -1 $O/bench_c2.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_executed'], d['stage_ms'], d.get('pair_setup_ms'), d['eval_mode']['ms_per_step'], d['gather_stage']['frac'], d['rank_share'], d['power'])"
for c in c3 c4 c5; do timeout 600 python bench.py --config $c --cpu-rays 0 > $O/bench_$c.log 2>&1; echo "bench $c rc=$?"; tail -1 $O/bench_$c.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_executed'], d['stage_ms'], d.get('pair_setup_ms'), d['eval_mode']['ms_per_step'], d['gather_stage']['frac'], d['lattice_bytes'], d['workspace_bytes'])"; done
timeout 900 bash tools/profile_bench.sh r6final2 > $O/profile.log 2>&1; echo "profile rc=$?"; tail -5 $O/profile.log
timeout 600 python tools/bench_variants.py > $O/variants.log 2>&1; echo "variants rc=$?"; grep -v amdgpu.ids $O/variants.log
timeout 300 python tools/train_step_probe.py > $O/train.log 2>&1; echo "train rc=$?"; tail -3 $O/train.log
timeout 300 python tools/bench_wgrad.py > $O/bench_wgrad.log 2>&1; grep -v amdgpu.ids $O/bench_wgrad.log
timeout 300 tools/_dev/wgrad_probe > $O/wgrad_probe.log 2>&1; cat $O/wgrad_probe.log
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trainprof -o train -- python $GRAFT_REPO_ROOT/tools/train_step_probe.py 10 > $GRAFT_REPO_ROOT/$O/trainprof.log 2>&1; cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3, re, glob
for f in glob.glob('gpurun_out/r6final2/trainprof/**/*_results.db', recursive=True):
    c = sqlite3.connect(f).cursor()
    rows = list(c.execute("select name, count(*), avg(duration), sum(duration) from kernels group by name order by 4 desc"))
    print("train trace: total ms", sum(r[3] for r in rows) / 1e6, "over 33 steps")
    for name, n, avg, s in rows[:32]:
        print(f"{s/1e6/33:8.3f} ms/step {n/33:6.1f} x {avg/1e3:8.1f} us  {re.sub(r'.anonymous namespace.::', '', name)[:100]}")
PY
find $O/trainprof -name "*.db" -size +3M -delete
