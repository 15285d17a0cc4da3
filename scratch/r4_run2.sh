#!/bin/bash
# round 4, GPU call 2: (a) bisection of the <double, 7, 0> miscompute, (b) polish {inline, call} x FRESH_LANE {off, on} on every
# instantiation with checksums against the round-3 build, (c) HBM traffic of the headline for the two polish forms
mkdir -p gpurun_out
AB=$PWD/racing-lmpc-ros2_amd/lib/ab
for lib in r3 rc_m01 rc_m08 rc_m10 rc_m09 rc_m18 rc_m11 rc_wait0 rc_nosgpr rc_nopost rc_O2; do
  LMPC_HIP_LIBRARY=$AB/liblmpc_$lib.so timeout 120 python scratch/r4_ab.py trk40 iac > gpurun_out/r4b_rc_$lib.jsonl 2> gpurun_out/r4b_rc_$lib.err
done
for lib in r3 inl_nf call_nf inl call; do
  LMPC_HIP_LIBRARY=$AB/liblmpc_$lib.so timeout 400 python scratch/r4_ab.py trk10 trk20 trk40 trk60 trk80 lmpc lmpc96 lmpc40 lmpc60 lmpc80 iac iac80 trk60m > gpurun_out/r4b_ab_$lib.jsonl 2> gpurun_out/r4b_ab_$lib.err
done
for lib in inl call; do
  LMPC_HIP_LIBRARY=$AB/liblmpc_$lib.so timeout 300 python bench.py --no-others --no-cpu-baseline --no-batch1 --steps 20 > gpurun_out/r4b_bench_$lib.json 2> gpurun_out/r4b_bench_$lib.err
done
timeout 300 python scratch/r4_ab.py iac lmpc32kreg trk20 > gpurun_out/r4b_ab_main.jsonl 2> gpurun_out/r4b_ab_main.err
ls gpurun_out | grep r4b | wc -l
