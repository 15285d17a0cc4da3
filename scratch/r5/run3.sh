#!/bin/bash
# round 5, GPU call 3: the dispatch sweep on both libraries with the failing cases dumped for CPU analysis
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
LMPC_HIP_LIBRARY=$PWD/racing-lmpc-ros2_amd/lib/liblmpc_hip.so python tests/dispatch_sweep.py --problems 1024 --dump gpurun_out/r5c_dump_product > gpurun_out/r5c_sweep_product.txt 2>&1
LMPC_HIP_LIBRARY=$PWD/racing-lmpc-ros2_amd/lib/liblmpc_hip_dbg.so python tests/dispatch_sweep.py --problems 1024 --dump gpurun_out/r5c_dump_dbg > gpurun_out/r5c_sweep_dbg.txt 2>&1
grep -c . gpurun_out/r5c_sweep_product.txt; grep "<--" gpurun_out/r5c_sweep_product.txt | cut -c1-200; echo; grep "<--" gpurun_out/r5c_sweep_dbg.txt | cut -c1-200 | head -50
