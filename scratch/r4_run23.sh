#!/bin/bash
# round 4, GPU call 23: the final build, every problem of every bench batch against the twin; full suite; default bench
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout 1500 python scratch/r4_fullsize_parity.py > gpurun_out/r4w_fullsize_parity.txt 2>&1
( time python -m pytest tests -x -q -m gpu 2>&1 | tail -2 ) > gpurun_out/r4w_pytest.log 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r4w_bench.json 2> gpurun_out/r4w_bench.err
grep -v amdgpu.ids gpurun_out/r4w_fullsize_parity.txt; cat gpurun_out/r4w_pytest.log; tail -1 gpurun_out/r4w_bench.json | head -c 600
