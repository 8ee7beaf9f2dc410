#!/bin/bash
# round 6: dry runs of the multi-rank bench path on ONE device (gloo; no scaling figure), config 5 on the unposed pair, its fixture on the GPU
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r6h
mkdir -p $O
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --backend gloo --device 0 --steps 3 --warmup 1 > $O/c2_2rank.log 2>&1; echo "c2 2-rank rc=$?"
tail -1 $O/c2_2rank.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['n_gpus'], round(d['ms_per_step'],2), d['config']['parallelism'][:80], d['frame_per_rank'] and round(d['frame_per_rank']['ms_per_step'],2))"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --backend gloo --device 0 --config c3 --rays-per-scene 8192 --steps 3 --warmup 1 > $O/c3_2rank.log 2>&1; echo "c3 2-rank rc=$?"
tail -1 $O/c3_2rank.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['n_gpus'], round(d['ms_per_step'],2), d['config']['workload'][:140], d['config']['rays_per_step'], d['config']['rays_per_step_per_gpu'])"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 8 --backend gloo --device 0 --config c3 --rays-per-scene 16384 --steps 2 --warmup 1 --no-extras > $O/c3_8rank.log 2>&1; echo "c3 8-rank rc=$?"
tail -1 $O/c3_8rank.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['n_gpus'], round(d['ms_per_step'],2), d['config']['rays_per_step'], d['config']['rays_per_step_per_gpu'])"
timeout 900 python bench.py --config c5 --cpu-rays 0 > $O/bench_c5.log 2>&1; echo "c5 rc=$?"
tail -1 $O/bench_c5.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5', round(d['ms_per_step'],2), d['value'], d['roofline']['frac'], d['config']['workload'][:120]); print(d['stage_ms'])"
timeout 900 python bench.py --config c3 --cpu-rays 0 > $O/bench_c3.log 2>&1; echo "c3 rc=$?"
tail -1 $O/bench_c3.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3', round(d['ms_per_step'],2), d['rank_share'])"
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_harness.py -m gpu -q -x -p no:cacheprovider -k "t2_c5 or bench_prints or c5" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
