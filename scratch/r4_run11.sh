#!/bin/bash
# round 4, GPU call 11: final build (DPP sweeps everywhere) -- GPU suite, smoke, default bench, the bench lines that moved, reproducibility
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
( time python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r4k_pytest.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4k_smoke.log 2>&1
O=gpurun_out
run() { name=$1; shift; python bench.py "$@" 2>$O/r04_bench_$name.err | tail -1 > $O/r04_bench_$name.json; }
run tracking
run tracking_s1 --streams 1 --no-others --no-cpu-baseline
run tracking_n40 --horizon 40 --no-others --no-cpu-baseline --steps 20
run lmpc --workload lmpc --no-others --no-cpu-baseline
run lmpc_b32768 --workload lmpc --batch 32768 --steps 10 --no-others --no-cpu-baseline
run iac_n40 --workload iac --horizon 40 --batch 8192 --steps 20 --no-others --no-cpu-baseline
run iac_n40_mixed --workload iac --horizon 40 --batch 8192 --precision mixed --steps 20 --no-others --no-cpu-baseline
run lmpc_b32768_mixed --workload lmpc --batch 32768 --precision mixed --steps 10 --no-others --no-cpu-baseline
run lmpc_b32768_mixed_regression --workload lmpc --batch 32768 --precision mixed --regression --steps 10 --no-others --no-cpu-baseline
bash scratch/prof.sh lmpc --workload lmpc > gpurun_out/prof_lmpc.log 2>&1
( bash scratch/r2_det_all.sh; python scratch/r3_det_mixed.py ) > gpurun_out/r4k_determinism.txt 2>&1
grep -E "passed|failed" gpurun_out/r4k_pytest.log; tail -1 gpurun_out/r4k_smoke.log
