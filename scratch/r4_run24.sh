#!/bin/bash
# round 4, GPU call 24: fused factor + predictor backward sweep against the separate sweep (scratch/r4_build_variants4.sh)
mkdir -p gpurun_out
AB=racing-lmpc-ros2_amd/lib/ab
run() { v=$1; shift; LMPC_HIP_LIBRARY=$PWD/$AB/liblmpc_$v.so timeout 600 python scratch/r4_ab.py "$@" 2>&1 | grep -E '^\{' ; }
{
for v in unfused fused; do run $v trk10 trk20 trk20big trk40 lmpc lmpc40 iac lmpc32kreg lmpc96; done
} > gpurun_out/r4x_ab.jsonl
python scratch/r4_pivot.py gpurun_out/r4x_ab.jsonl | tee gpurun_out/r4x_pivot.md
