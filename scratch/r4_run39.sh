#!/bin/bash
# round 4, GPU call 39 (final state of the round): + s_setprio around the serial stage chains of the two-waves-per-SIMD kernels, k-NN loads batched, regression loads grouped -- suite, smoke,
# determinism, bench lines, tracking profile, default bench
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
( time python -m pytest tests -x -q -m gpu 2>&1 | tail -2 ) > gpurun_out/r4af_pytest.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4af_smoke.log 2>&1
( bash scratch/r2_det_all.sh; python scratch/r3_det_mixed.py ) > gpurun_out/r4af_det.txt 2>&1
bash scratch/r4_bench_lines.sh > gpurun_out/r4af_bench_lines.txt 2>&1
bash scratch/prof.sh tracking > gpurun_out/prof_tracking.log 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r4af_bench.json 2> gpurun_out/r4af_bench.err
cat gpurun_out/r4af_pytest.log; tail -1 gpurun_out/r4af_smoke.log; grep -c "diffs vs rep 0 0\|(summed) 0\|first: 0 of 5" gpurun_out/r4af_det.txt; grep -v "diffs vs rep 0 0\|(summed) 0\|first: 0 of 5" gpurun_out/r4af_det.txt | head; cat gpurun_out/r4af_bench_lines.txt | cut -c1-260; head -6 gpurun_out/prof_tracking/summary.md; tail -1 gpurun_out/r4af_bench.json | head -c 400
