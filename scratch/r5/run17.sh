#!/bin/bash
# round 5, GPU call 19: the suite and the soak on the knot-parallel loop tail
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
( time python -m pytest tests -q -m gpu ) > gpurun_out/r05_gpu_suite.txt 2>&1
python scratch/r5/soak_fused.py 2>&1 | grep -v amdgpu > gpurun_out/r05_closed_loop_fused.txt
tail -25 gpurun_out/r05_gpu_suite.txt; cat gpurun_out/r05_closed_loop_fused.txt
