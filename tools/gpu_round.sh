set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --maxfail=40 --timeout 900 > gpurun_out/r2_pytest2.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r2_pytest2.log
timeout 600 python bench.py > gpurun_out/r2_bench2.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/r2_bench2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','stage_ms')}); print(d['gather_stage'])"
