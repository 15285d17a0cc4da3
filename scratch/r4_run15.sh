#!/bin/bash
mkdir -p gpurun_out
AB=$PWD/racing-lmpc-ros2_amd/lib/ab
python scratch/r4_tail32k.py > gpurun_out/r4o_tail_main.jsonl 2> gpurun_out/r4o_tail_main.err
SEED=1 python scratch/r4_tail32k.py > gpurun_out/r4o_tail_main_seed1.jsonl 2> /dev/null
SEED=2 python scratch/r4_tail32k.py > gpurun_out/r4o_tail_main_seed2.jsonl 2> /dev/null
for lib in stepC feasC dualC; do
LMPC_HIP_LIBRARY=$AB/liblmpc_$lib.so python scratch/r4_tail32k.py > gpurun_out/r4o_tail_$lib.jsonl 2> gpurun_out/r4o_tail_$lib.err
done
cat gpurun_out/r4o_tail_*.jsonl
