cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# multi-rank code path of bench.py, two ranks on the one GPU of this box over gloo (RCCL refuses two ranks on one device)
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 1 --cpu-rays 0 --backend gloo --device 0 > gpurun_out/r2_bench_2rank_gloo.log 2>&1; echo "2-rank rc=$?"
grep -v amdgpu gpurun_out/r2_bench_2rank_gloo.log | tail -5 | cut -c1-1500
