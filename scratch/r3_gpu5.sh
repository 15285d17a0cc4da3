#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/r3_time5.log
LMPC_LIB=$PWD/racing-lmpc-ros2_amd/lib/liblmpc_hip_nopolish.so timeout 300 python scratch/r3_time.py trk20 lmpc iac 2>&1 | grep -v "amdgpu.ids\|polish=on\|one pass" | sed 's/^/NOPOLISH-BUILD /' >> gpurun_out/r3_time5.log
timeout 300 python scratch/r3_time.py trk20 lmpc iac 2>&1 | grep -v amdgpu.ids | sed 's/^/V4          /' >> gpurun_out/r3_time5.log
timeout 300 python scratch/r3_diag.py > gpurun_out/r3_diag5.log 2>&1
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r3_pytest5.log
cut -c1-250 gpurun_out/r3_time5.log; tail -22 gpurun_out/r3_diag5.log; tail -12 gpurun_out/r3_pytest5.log
