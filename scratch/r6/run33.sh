cd $GRAFT_REPO_ROOT
C=lmpc:40:f64,lmpc:60:f64
timeout 600 python scratch/r6/fuse_check.py plain $C 2>&1 | grep -v amdgpu.ids
LMPC_HIP_LIBRARY=racing-lmpc-ros2_amd/lib/liblmpc_hip_f5f.so timeout 600 python scratch/r6/fuse_check.py fused $C 2>&1 | grep -v amdgpu.ids
python scratch/r6/fuse_check.py cmp plain fused
rm -f gpurun_out/fuse_fused.npz gpurun_out/fuse_plain.npz
