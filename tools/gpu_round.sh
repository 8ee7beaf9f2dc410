#!/bin/bash
# What the driver runs at round end, in one call: the GPU test suite, smoke(), the default bench line.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --maxfail=10 --timeout 900 -p no:cacheprovider > gpurun_out/final_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/final_pytest.log | tail -5
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/final_smoke.log 2>&1; echo "smoke rc=$?"; grep -E "smoke|hipcc" gpurun_out/final_smoke.log
timeout 900 python bench.py > gpurun_out/final_bench.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/final_bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['gather_stage']['frac'], d['cpu_baseline']['value'])"
