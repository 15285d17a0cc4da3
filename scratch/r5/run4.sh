#!/bin/bash
# round 5, GPU call 4: MA_MAX = 6 + the noise rule: dispatch sweep (both libraries, dumps), whole suite, every bench line
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
LMPC_HIP_LIBRARY=$PWD/racing-lmpc-ros2_amd/lib/liblmpc_hip.so python tests/dispatch_sweep.py --problems 1024 --dump gpurun_out/r5d_dump_product > gpurun_out/r5d_sweep_product.txt 2>&1
( time python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -40 ) > gpurun_out/r5d_pytest.log 2>&1
bash scratch/r5/bench_lines.sh > gpurun_out/r5d_bench_lines.txt 2>&1
grep -c . gpurun_out/r5d_sweep_product.txt; grep "<--" gpurun_out/r5d_sweep_product.txt | cut -c1-260; tail -1 gpurun_out/r5d_sweep_product.txt | cut -c1-600; cat gpurun_out/r5d_pytest.log; cut -c1-330 gpurun_out/r5d_bench_lines.txt
