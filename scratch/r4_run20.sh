#!/bin/bash
# round 4, GPU call 20: the row-phase build (slots per chunk, opaque slot tables, fresh lane in the model stream) -- suite, smoke,
# determinism, every bench line, profiles of the N = 60 and the configs[4] workloads
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
( time python -m pytest tests -x -q -m gpu -s 2>&1 | grep -E "passed|failed|configs\[|Error" ) > gpurun_out/r4t_pytest.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4t_smoke.log 2>&1
python scratch/r3_det_mixed.py > gpurun_out/r4t_det_mixed.txt 2>&1
bash scratch/r2_det_all.sh > gpurun_out/r4t_det.txt 2>&1
bash scratch/r4_bench_lines.sh > gpurun_out/r4t_bench_lines.txt 2>&1
bash scratch/prof.sh n60 --horizon 60 > gpurun_out/prof_n60.log 2>&1
bash scratch/prof.sh lmpcmix --workload lmpc --batch 32768 --precision mixed --regression > gpurun_out/prof_lmpcmix.log 2>&1
cat gpurun_out/r4t_pytest.log; tail -1 gpurun_out/r4t_smoke.log; tail -4 gpurun_out/r4t_det_mixed.txt; cat gpurun_out/r4t_det.txt; cat gpurun_out/r4t_bench_lines.txt
