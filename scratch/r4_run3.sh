#!/bin/bash
# round 4, GPU call 3: the whole GPU suite on the build with the measured policy, the per-problem look at what the
# miscomputing <double, 7, 0> build gets wrong, the default bench line
mkdir -p gpurun_out
AB=$PWD/racing-lmpc-ros2_amd/lib/ab
( time python -m pytest tests -m gpu -q -s 2>&1 | tail -60 ) > gpurun_out/r4c_pytest.log 2>&1
for lib in r3 rc_m09 inl_nf call; do
  LMPC_HIP_LIBRARY=$AB/liblmpc_$lib.so timeout 200 python scratch/r4_cmp_builds.py save $lib 2> gpurun_out/r4c_cmp_$lib.err
done
python scratch/r4_cmp_builds.py diff r3 rc_m09 > gpurun_out/r4c_cmp_r3_vs_m09.jsonl 2>&1
python scratch/r4_cmp_builds.py diff r3 inl_nf > gpurun_out/r4c_cmp_r3_vs_inl_nf.jsonl 2>&1
python scratch/r4_cmp_builds.py diff r3 call > gpurun_out/r4c_cmp_r3_vs_call.jsonl 2>&1
timeout 300 python scratch/r4_ab.py trk10 trk20 trk40 trk60 trk80 lmpc lmpc96 lmpc40 lmpc60 iac iac80 lmpc32kreg > gpurun_out/r4c_ab_main.jsonl 2> gpurun_out/r4c_ab_main.err
( time python bench.py ) > gpurun_out/r4c_bench.json 2> gpurun_out/r4c_bench.err
tail -4 gpurun_out/r4c_pytest.log
