#!/bin/bash
# round 4, GPU call 1: the GPU suite on the new build, A/B timing of the builds under racing-lmpc-ros2_amd/lib/ab,
# the fp32 acceptance-test variants on the two reduced-precision configs, the default bench line
mkdir -p gpurun_out
AB=racing-lmpc-ros2_amd/lib/ab
( time python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fullsize.py 2>&1 | tail -15 ) > gpurun_out/r4a_pytest.log 2>&1
( time python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s 2>&1 | tail -40 ) > gpurun_out/r4a_fullsize.log 2>&1
for lib in r3 inl call nofresh; do
  LMPC_HIP_LIBRARY=$PWD/$AB/liblmpc_$lib.so timeout 300 python scratch/r4_ab.py trk20 trk10 lmpc lmpc32kreg iac trk40 trk60 > gpurun_out/r4a_ab_$lib.jsonl 2> gpurun_out/r4a_ab_$lib.err
done
timeout 300 python scratch/r4_ab.py trk20 trk10 lmpc lmpc32kreg iac trk40 trk60 trk20big > gpurun_out/r4a_ab_main.jsonl 2> gpurun_out/r4a_ab_main.err
for lib in tolA tolB tolC tolD; do
  LMPC_HIP_LIBRARY=$PWD/$AB/liblmpc_$lib.so timeout 200 python scratch/r4_ab.py lmpc32kreg iac lmpc > gpurun_out/r4a_ab_$lib.jsonl 2> gpurun_out/r4a_ab_$lib.err
done
for lib in mll mllD; do
  LMPC_HIP_LIBRARY=$PWD/$AB/liblmpc_$lib.so timeout 200 python scratch/r4_ab.py lmpc40 lmpc60 > gpurun_out/r4a_ab_$lib.jsonl 2> gpurun_out/r4a_ab_$lib.err
done
( time python bench.py ) > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err
tail -3 gpurun_out/r4a_pytest.log; tail -5 gpurun_out/r4a_fullsize.log
