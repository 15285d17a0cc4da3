#!/bin/bash
# round 5, GPU call 12: warm-start rounds against the closed loop; the full GPU suite on the final build; smoke; default bench; the 3-lap learning line
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
python scratch/r5/soak_warm_rounds.py 2>&1 | grep -v amdgpu > gpurun_out/r05_closed_loop_warm_rounds.txt
( time python -m pytest tests -q -m gpu --durations=12 ) > gpurun_out/r05_gpu_suite.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.txt 2>&1; echo "smoke rc $?" >> gpurun_out/r05_smoke.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
python bench.py --workload lmpc --laps 3 --no-others --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_bench_lmpc_laps3.json
python bench.py --workload lmpc --laps 3 --lmpc-data near --no-others --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 > gpurun_out/r05_bench_lmpc_laps3_near.json
cat gpurun_out/r05_closed_loop_warm_rounds.txt; tail -25 gpurun_out/r05_gpu_suite.txt; tail -3 gpurun_out/r05_smoke.txt; tail -4 gpurun_out/r05_bench_default.err; cut -c1-400 gpurun_out/r05_bench_lmpc_laps3.json; echo; cut -c1-300 gpurun_out/r05_bench_lmpc_laps3_near.json
