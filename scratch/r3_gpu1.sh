#!/bin/bash
# first GPU call of round 3: the suite, then the timing / accuracy sweep with the polish on and off
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_mixed_lmpc.py 2>&1 | tail -40 > gpurun_out/r3_pytest1.log
timeout 600 python scratch/r3_time.py > gpurun_out/r3_time1.log 2>&1
timeout 300 python -m pytest tests/test_gpu_mixed_lmpc.py -m gpu -q 2>&1 | tail -30 > gpurun_out/r3_pytest1b.log
tail -5 gpurun_out/r3_pytest1.log; cat gpurun_out/r3_time1.log
