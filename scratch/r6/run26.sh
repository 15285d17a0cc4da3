cd $GRAFT_REPO_ROOT
timeout 600 python scratch/r6/w2_check.py 4096 41,44,48,52,56,60,64,65,72 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_w2_threshold.txt
timeout 3000 python -m pytest tests -m gpu -q -x > gpurun_out/r06_gpu_suite.txt 2>&1; echo "rc $?" >> gpurun_out/r06_gpu_suite.txt
tail -n 6 gpurun_out/r06_gpu_suite.txt
