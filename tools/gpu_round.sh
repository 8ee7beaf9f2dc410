set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --maxfail=40 --timeout 900 > gpurun_out/r2_pytest3.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r2_pytest3.log
