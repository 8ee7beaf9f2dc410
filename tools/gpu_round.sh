cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_bench.sh round2 > gpurun_out/profile_round2.log 2>&1; echo "profile rc=$?"
cat gpurun_out/profile_round2.log | tail -8
bash tools/pmc_fused.sh 0 > gpurun_out/pmc_fused_round2.log 2>&1; echo "pmc rc=$?"
tail -40 gpurun_out/pmc_fused_round2.log
head -40 gpurun_out/prof_round2_summary.md
