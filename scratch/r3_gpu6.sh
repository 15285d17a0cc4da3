#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/r3_pytest6.log
timeout 300 python scratch/r3_time.py trk20 lmpc iac 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_time6.log
tail -40 gpurun_out/r3_pytest6.log; cut -c1-260 gpurun_out/r3_time6.log
