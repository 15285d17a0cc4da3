#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scratch/prof.sh tracking > gpurun_out/prof_tracking.log 2>&1
bash scratch/prof.sh lmpc --workload lmpc > gpurun_out/prof_lmpc.log 2>&1
bash scratch/prof.sh n60 --horizon 60 > gpurun_out/prof_n60.log 2>&1
bash scratch/prof.sh lmpcmix --workload lmpc --batch 32768 --precision mixed --regression > gpurun_out/prof_lmpcmix.log 2>&1
for f in gpurun_out/prof_*.log; do tail -n 5 $f; done
