#!/bin/bash
# round 4, GPU call 4: row-state dump of the miscomputing build, rocprofv3 trace + counters of the final build, every bench line,
# bitwise reproducibility of every kernel family
mkdir -p gpurun_out
AB=$PWD/racing-lmpc-ros2_amd/lib/ab
for lib in dump_m01 dump_m09; do
  LMPC_HIP_LIBRARY=$AB/liblmpc_$lib.so timeout 200 python scratch/r4_rowdump.py save $lib 2> gpurun_out/r4d_rowdump_$lib.err
done
python scratch/r4_rowdump.py diff dump_m01 dump_m09 > gpurun_out/r4d_rowdump.jsonl 2>&1
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
bash scratch/prof.sh tracking > gpurun_out/prof_tracking.log 2>&1
bash scratch/prof.sh lmpc --workload lmpc > gpurun_out/prof_lmpc.log 2>&1
bash scratch/prof.sh n60 --horizon 60 > gpurun_out/prof_n60.log 2>&1
bash scratch/prof.sh lmpcmix --workload lmpc --batch 32768 --precision mixed --regression > gpurun_out/prof_lmpcmix.log 2>&1
bash scratch/prof.sh iacf32 --workload iac --horizon 40 --batch 8192 --precision f32 > gpurun_out/prof_iacf32.log 2>&1
bash scratch/r4_bench_lines.sh > gpurun_out/r4d_bench_lines.log 2>&1
( bash scratch/r2_det_all.sh; python scratch/r3_det_mixed.py ) > gpurun_out/r4d_determinism.txt 2>&1
tail -3 gpurun_out/r4d_determinism.txt; tail -15 gpurun_out/r4d_bench_lines.log
