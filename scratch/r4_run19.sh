#!/bin/bash
# round 4, GPU call 19: row-phase policies per instantiation (scratch/r4_build_variants3.sh)
mkdir -p gpurun_out
AB=racing-lmpc-ros2_amd/lib/ab
run() { v=$1; shift; LMPC_HIP_LIBRARY=$PWD/$AB/liblmpc_$v.so timeout 600 python scratch/r4_ab.py "$@" 2>&1 | grep -E '^\{' ; }
{
run old trk20 trk40 trk60 trk80 lmpc lmpc40 lmpc60 lmpc80 iac iac60 iac80 lmpc32kreg lmpc32k lmpc96 trk10
run p1 trk20 trk40 trk60 trk80 lmpc lmpc40 lmpc60 lmpc80 iac iac60 iac80 lmpc32kreg lmpc32k lmpc96 trk10
for v in p1_lc p1_c4 p1_c8; do run $v trk60 trk80 iac80 lmpc60 lmpc80; done
} > gpurun_out/r4s_ab.jsonl
python scratch/r4_pivot.py gpurun_out/r4s_ab.jsonl | tee gpurun_out/r4s_pivot.md
