#!/bin/bash
# round 3, GPU call 12: the round's closing measurements — GPU suite, bench line, rocprofv3 trace + HBM PMC passes, fused-kernel counters,
# an 8-rank dry run of the banded bench on the one GPU (gloo)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c12
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 --timeout 900 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest.log | tail -8
timeout 900 python bench.py > $OUT/bench.log 2>&1; echo "bench rc=$?"; tail -1 $OUT/bench.log | cut -c1-400
timeout 900 python bench.py --chunk-rays 8192 --cpu-rays 0 --no-extras > $OUT/bench_chunk8192.log 2>&1; echo "bench chunked rc=$?"; tail -1 $OUT/bench_chunk8192.log | cut -c1-300
bash tools/profile_bench.sh r3 > $OUT/profile.log 2>&1; echo "profile rc=$?"; tail -5 $OUT/profile.log
bash tools/pmc_fused.sh 0 > $OUT/pmc_fused.log 2>&1; echo "pmc rc=$?"; grep -E "avg=" $OUT/pmc_fused.log | head -40
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 8 --backend gloo --device 0 --steps 3 --warmup 1 --cpu-rays 0 --no-extras > $OUT/bench_8rank_gloo.log 2>&1; echo "8-rank rc=$?"; tail -1 $OUT/bench_8rank_gloo.log | cut -c1-1200
