cd $GRAFT_REPO_ROOT
echo "== lean (shipped)"; timeout 600 python scratch/r6/w2_check.py 4096 41,42,48,53 2>&1 | grep -v amdgpu.ids
echo "== fat everywhere"; LMPC_HIP_LIBRARY=racing-lmpc-ros2_amd/lib/liblmpc_hip_fat.so timeout 600 python scratch/r6/w2_check.py 4096 41,42,48,53 2>&1 | grep -v amdgpu.ids
