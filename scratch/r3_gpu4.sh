#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/r3_time4.log
LMPC_LIB=$PWD/racing-lmpc-ros2_amd/lib/liblmpc_hip_nopolish.so timeout 300 python scratch/r3_time.py trk20 trk40 lmpc iac 2>&1 | grep -v amdgpu.ids | sed 's/^/NOPOLISH-BUILD /' >> gpurun_out/r3_time4.log
timeout 300 python scratch/r3_time.py trk20 trk40 lmpc iac 2>&1 | grep -v amdgpu.ids | sed 's/^/V3          /' >> gpurun_out/r3_time4.log
cut -c1-140 gpurun_out/r3_time4.log
