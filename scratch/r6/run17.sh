# round 6, GPU call 17: fused factorisation A/B, then the GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python scratch/r6/fuse_check.py fused 2>&1 | grep -v amdgpu.ids
LMPC_HIP_LIBRARY=racing-lmpc-ros2_amd/lib/liblmpc_hip_nofuse.so timeout 900 python scratch/r6/fuse_check.py plain 2>&1 | grep -v amdgpu.ids
python scratch/r6/fuse_check.py cmp plain fused | tee gpurun_out/r6_fuse_cmp.txt
rm -f gpurun_out/fuse_fused.npz gpurun_out/fuse_plain.npz
timeout 3000 python -m pytest tests -m gpu -q -x > gpurun_out/r06_gpu_suite.txt 2>&1; echo "rc $?" >> gpurun_out/r06_gpu_suite.txt
tail -n 15 gpurun_out/r06_gpu_suite.txt
