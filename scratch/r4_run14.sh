#!/bin/bash
# round 4, GPU call 14: the two-translation-unit build (mixed learning kernels with the minimum-register scheduler)
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
( time python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r4n_pytest.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4n_smoke.log 2>&1
timeout 300 python scratch/r4_ab.py lmpc lmpc96 lmpc32kreg lmpc32k > gpurun_out/r4n_ab_main.jsonl 2> gpurun_out/r4n_ab_main.err
O=gpurun_out
run() { name=$1; shift; python bench.py "$@" 2>$O/r04_bench_$name.err | tail -1 > $O/r04_bench_$name.json; }
run lmpc_b32768_mixed --workload lmpc --batch 32768 --precision mixed --steps 10 --no-others --no-cpu-baseline
run lmpc_b32768_mixed_regression --workload lmpc --batch 32768 --precision mixed --regression --steps 10 --no-others --no-cpu-baseline
bash scratch/prof.sh lmpcmix --workload lmpc --batch 32768 --precision mixed --regression > gpurun_out/prof_lmpcmix.log 2>&1
python scratch/r3_det_mixed.py > gpurun_out/r4n_det_mixed.txt 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r4n_bench.json 2> gpurun_out/r4n_bench.err
grep -E "passed|failed" gpurun_out/r4n_pytest.log; tail -1 gpurun_out/r4n_smoke.log; cat gpurun_out/r4n_det_mixed.txt | tail -4
