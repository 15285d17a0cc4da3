#!/bin/bash
# round 4, GPU call 21: per-phase cycle split after the row-phase changes (profiling build)
mkdir -p gpurun_out
python scratch/phase_timing.py 60 4096 > gpurun_out/r4u_phase_n60.txt 2>&1
python scratch/phase_timing.py 80 4096 > gpurun_out/r4u_phase_n80.txt 2>&1
python scratch/phase_timing.py 20 4096 > gpurun_out/r4u_phase_n20.txt 2>&1
python scratch/phase_timing.py 20 32768 lmpc mixed > gpurun_out/r4u_phase_lmpcmix.txt 2>&1
python scratch/phase_timing.py 60 4096 lmpc > gpurun_out/r4u_phase_lmpc60.txt 2>&1
head -22 gpurun_out/r4u_phase_n60.txt; head -22 gpurun_out/r4u_phase_lmpcmix.txt
