#!/bin/bash
# round 5, first GPU call: full GPU suite on the partial-sum build, default bench, tail probe, power-sampling availability
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r5a
O=gpurun_out/r5a
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.log | tail -20
timeout 600 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"
tail -1 $O/bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['gather_stage']['frac'], d['stage_ms'], d.get('pair_setup_ms'), d.get('eval_mode'))"
timeout 300 python tools/tail_probe.py > $O/tail_probe.log 2>&1; echo "tail_probe rc=$?"; cat $O/tail_probe.log | tail -8
(which rocm-smi amd-smi; timeout 20 rocm-smi --showpower --showclocks 2>&1 | head -40; timeout 20 amd-smi metric -p 2>&1 | head -30; python -c "import amdsmi; print('amdsmi ok')" 2>&1 | tail -1; ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>&1 | head -40; cat /sys/class/drm/card*/device/hwmon/hwmon*/power1_average 2>&1 | head) > $O/smi.log 2>&1
tail -30 $O/smi.log
