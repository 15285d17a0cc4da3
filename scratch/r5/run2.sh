#!/bin/bash
# round 5, GPU call 2: dispatch sweep smoke, then the whole suite with the new tests (shipped horizons, dense fixtures, dispatch, facade)
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
( time python tests/dispatch_sweep.py --problems 256 --nstep 13 ) > gpurun_out/r5b_sweep_quick.txt 2>&1
racing-lmpc-ros2_amd/lib/test_racing_lmpc tests/golden/barc_track/15_barc_optm.txt > gpurun_out/r5b_racing_lmpc.txt 2>&1
( time python -m pytest tests -q -m gpu --durations=25 2>&1 | tail -70 ) > gpurun_out/r5b_pytest.log 2>&1
tail -30 gpurun_out/r5b_sweep_quick.txt; tail -8 gpurun_out/r5b_racing_lmpc.txt; cat gpurun_out/r5b_pytest.log
