#!/bin/bash
# round 5, GPU call 15: the fused loop tail -- its tests, the whole suite, the soak with and without it
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
python -m pytest tests/test_gpu_loop.py -q -x 2>&1 | tail -30 > gpurun_out/loop_tests.txt
python scratch/r5/soak_fused.py 2>&1 | grep -v amdgpu > gpurun_out/r05_closed_loop_fused.txt
( time python -m pytest tests -q -m gpu ) > gpurun_out/r05_gpu_suite.txt 2>&1
cat gpurun_out/loop_tests.txt; cat gpurun_out/r05_closed_loop_fused.txt; tail -6 gpurun_out/r05_gpu_suite.txt
