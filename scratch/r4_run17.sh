#!/bin/bash
# round 4, GPU call 17: final build (fp32 acceptance: rows 3e-6, multipliers -3e-5) -- suite, smoke, bench, configs[4] lines, profile
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
( time python -m pytest tests -x -q -m gpu -s 2>&1 | grep -E "passed|failed|configs\[|Error" ) > gpurun_out/r4q_pytest.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4q_smoke.log 2>&1
O=gpurun_out
run() { name=$1; shift; python bench.py "$@" 2>$O/r04_bench_$name.err | tail -1 > $O/r04_bench_$name.json; }
run lmpc_b32768_mixed --workload lmpc --batch 32768 --precision mixed --steps 10 --no-others --no-cpu-baseline
run lmpc_b32768_mixed_regression --workload lmpc --batch 32768 --precision mixed --regression --steps 10 --no-others --no-cpu-baseline
run iac_n40_mixed --workload iac --horizon 40 --batch 8192 --precision mixed --steps 20 --no-others --no-cpu-baseline
run iac_n40_f32 --workload iac --horizon 40 --batch 8192 --precision f32 --steps 20 --no-others --no-cpu-baseline
bash scratch/prof.sh lmpcmix --workload lmpc --batch 32768 --precision mixed --regression > gpurun_out/prof_lmpcmix.log 2>&1
python scratch/r3_det_mixed.py > gpurun_out/r4q_det_mixed.txt 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r4q_bench.json 2> gpurun_out/r4q_bench.err
cat gpurun_out/r4q_pytest.log; tail -1 gpurun_out/r4q_smoke.log; tail -4 gpurun_out/r4q_det_mixed.txt
