#!/bin/bash
# round 5, GPU call 10: warm tests again, soak with longest-first, B = 16384 soak
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
( time python -m pytest tests/test_gpu_warm.py -q -m gpu -s 2>&1 | grep -v "^$" | tail -12 ) > gpurun_out/r5j_pytest.log 2>&1
python scratch/r5/soak_warm.py > gpurun_out/r5j_soak_warm.txt 2>&1
cat gpurun_out/r5j_pytest.log; grep -v amdgpu gpurun_out/r5j_soak_warm.txt
