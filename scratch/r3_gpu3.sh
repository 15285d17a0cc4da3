#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python scratch/r3_diag.py > gpurun_out/r3_diag3.log 2>&1
for rep in 1 2; do
LMPC_LIB=$PWD/racing-lmpc-ros2_amd/lib/liblmpc_hip_nopolish.so timeout 300 python scratch/r3_time.py trk20 trk40 lmpc iac 2>&1 | grep -v amdgpu.ids | sed 's/^/NOPOLISH-BUILD /' >> gpurun_out/r3_time3.log
timeout 300 python scratch/r3_time.py trk20 trk40 lmpc iac 2>&1 | grep -v amdgpu.ids | sed 's/^/V2          /' >> gpurun_out/r3_time3.log
done
cat gpurun_out/r3_diag3.log; cut -c1-150 gpurun_out/r3_time3.log
