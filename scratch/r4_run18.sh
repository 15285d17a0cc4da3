#!/bin/bash
# round 4, GPU call 18: A/B of the row-phase variants (scratch/r4_build_variants2.sh) -- kernel ms + checksum per case
mkdir -p gpurun_out
AB=racing-lmpc-ros2_amd/lib/ab
run() { v=$1; shift; LMPC_HIP_LIBRARY=$PWD/$AB/liblmpc_$v.so timeout 600 python scratch/r4_ab.py "$@" 2>&1 | grep -E '^\{' ; }
{
run old trk20 trk40 trk60 trk80 lmpc lmpc40 lmpc60 lmpc80 iac iac60 iac80 lmpc32kreg
run new trk60 trk80 lmpc60 lmpc80 iac60 iac80
for v in c3 c6 nofetch newnc; do run $v trk60 iac80 lmpc60; done
run op7 trk40 lmpc40 iac iac60
run op7c trk40 lmpc40 iac
run op4 trk20 lmpc lmpc32kreg iac trk40
} > gpurun_out/r4r_ab.jsonl
python scratch/r4_pivot.py gpurun_out/r4r_ab.jsonl | tee gpurun_out/r4r_pivot.md
