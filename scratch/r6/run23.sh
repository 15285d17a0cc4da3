cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_suite.txt 2>&1; echo "rc $?" >> gpurun_out/r06_gpu_suite.txt
tail -n 40 gpurun_out/r06_gpu_suite.txt
