cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cam in host device host device; do
timeout 600 python bench.py --cpu-rays 0 --no-extras --cameras $cam > gpurun_out/r2_bench_cam_$cam.log 2>&1; echo "bench rc=$?"
grep -v amdgpu gpurun_out/r2_bench_cam_$cam.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cam', d['value'], d['ms_per_step'], sum(d['stage_ms'].values()))"
done
