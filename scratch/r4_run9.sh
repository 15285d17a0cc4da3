#!/bin/bash
# round 4, GPU call 9: after the DPP lean solve -- GPU suite, N = 60 / 80 profile and bench lines, reproducibility
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
( time python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r4i_pytest.log 2>&1
bash scratch/prof.sh n60 --horizon 60 > gpurun_out/prof_n60.log 2>&1
O=gpurun_out
run() { name=$1; shift; python bench.py "$@" 2>$O/r04_bench_$name.err | tail -1 > $O/r04_bench_$name.json; }
run tracking_n60 --horizon 60 --no-others --no-cpu-baseline --steps 10
run tracking_n80 --horizon 80 --no-others --no-cpu-baseline --steps 10
run tracking_n40 --horizon 40 --no-others --no-cpu-baseline --steps 20
( bash scratch/r2_det_all.sh; python scratch/r3_det_mixed.py ) > gpurun_out/r4i_determinism.txt 2>&1
timeout 300 python scratch/r4_ab.py lmpc40 lmpc60 lmpc80 trk60 trk80 iac80 > gpurun_out/r4i_ab_main.jsonl 2> gpurun_out/r4i_ab_main.err
tail -3 gpurun_out/r4i_pytest.log; grep -c "differ" gpurun_out/r4i_determinism.txt
