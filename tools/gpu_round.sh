cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python tools/validate_frame.py 0.5 0.1 2>/dev/null | grep -v amdgpu > gpurun_out/round2_whole_frame_parity.md; echo "validate rc=$?"
cat gpurun_out/round2_whole_frame_parity.md
