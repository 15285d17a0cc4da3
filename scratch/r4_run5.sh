#!/bin/bash
mkdir -p gpurun_out
AB=$PWD/racing-lmpc-ros2_amd/lib/ab
for lib in r3 inl call; do
  LMPC_HIP_LIBRARY=$AB/liblmpc_$lib.so timeout 300 python scratch/r4_backtoback.py > gpurun_out/r4e_b2b_$lib.jsonl 2> gpurun_out/r4e_b2b_$lib.err
done
timeout 300 python scratch/r4_backtoback.py > gpurun_out/r4e_b2b_main.jsonl 2> gpurun_out/r4e_b2b_main.err
cat gpurun_out/r4e_b2b_*.jsonl
