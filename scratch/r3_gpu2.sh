#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python scratch/r3_time.py > gpurun_out/r3_time2.log 2>&1
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/r3_pytest2.log
cat gpurun_out/r3_time2.log; tail -15 gpurun_out/r3_pytest2.log
