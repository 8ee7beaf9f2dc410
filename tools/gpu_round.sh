cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_bench.sh round2 > gpurun_out/profile_round2.log 2>&1; echo "profile rc=$?"
timeout 600 python bench.py > gpurun_out/r2_bench_final.log 2>&1; echo "bench rc=$?"
grep -v amdgpu gpurun_out/r2_bench_final.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'], d['gather_stage']['frac'], d['roofline']['frac'])"
head -26 gpurun_out/prof_round2_summary.md
