#!/bin/bash
# round 5, GPU call 1: the polish's step rule (up to 4 steps, 1e-7 / 1e-6, 4 rounds) + RacingLMPC facade + sharded solver:
# suite, smoke, every bench batch against the twin, default bench
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
( time python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r5a_pytest.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5a_smoke.log 2>&1
python scratch/r4_fullsize_parity.py > gpurun_out/r5a_fullsize_parity.txt 2>&1
racing-lmpc-ros2_amd/lib/bench_cabi tests/golden/barc_track/15_barc_optm.txt 4096 200 > gpurun_out/r5a_cabi.txt 2>&1
racing-lmpc-ros2_amd/lib/bench_cabi tests/golden/barc_track/15_barc_optm.txt 4096 200 --gpus 1 --same-device >> gpurun_out/r5a_cabi.txt 2>&1
racing-lmpc-ros2_amd/lib/bench_cabi tests/golden/barc_track/15_barc_optm.txt 4096 200 --gpus 2 --same-device >> gpurun_out/r5a_cabi.txt 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err
cat gpurun_out/r5a_pytest.log; tail -1 gpurun_out/r5a_smoke.log; cat gpurun_out/r5a_fullsize_parity.txt; cat gpurun_out/r5a_cabi.txt; tail -1 gpurun_out/r5a_bench.json | head -c 600
