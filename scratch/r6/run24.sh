cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_suite.txt 2>&1; echo "rc $?" >> gpurun_out/r06_gpu_suite.txt
tail -n 12 gpurun_out/r06_gpu_suite.txt
timeout 600 python scratch/r6/w2_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_w2_check_fused.txt
