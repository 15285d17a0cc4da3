cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sweep_dump
timeout 1700 python tests/dispatch_sweep.py --problems 1024 --dump gpurun_out/sweep_dump > gpurun_out/r6_sweep_product.txt 2>&1; echo "sweep rc $?"
grep -E "<--|fault|library" gpurun_out/r6_sweep_product.txt | head -30
LMPC_HIP_LIBRARY=racing-lmpc-ros2_amd/lib/liblmpc_hip_nofuse.so timeout 300 python scratch/r6/fuse_check.py plain 2>&1 | grep -v amdgpu.ids > /dev/null
timeout 300 python scratch/r6/fuse_check.py fused 2>&1 | grep -v amdgpu.ids > /dev/null
python scratch/r6/fuse_check.py cmp plain fused | tee gpurun_out/r6_fuse_cmp2.txt
rm -f gpurun_out/fuse_fused.npz gpurun_out/fuse_plain.npz
