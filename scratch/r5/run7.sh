#!/bin/bash
# round 5, GPU call 8: dedupe + noise rule + index mode: whole suite (no -x), determinism scripts, closed-loop soak
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
( time python -m pytest tests -q -m gpu --durations=5 2>&1 | tail -60 ) > gpurun_out/r5h_pytest.log 2>&1
( bash scratch/r2_det_all.sh; python scratch/r3_det_mixed.py ) > gpurun_out/r5h_det.txt 2>&1
python scratch/soak_graph.py > gpurun_out/r5h_soak.txt 2>&1
cat gpurun_out/r5h_pytest.log; grep -c "diffs vs rep 0 0\|(summed) 0\|first: 0 of 5" gpurun_out/r5h_det.txt; grep -v "diffs vs rep 0 0\|(summed) 0\|first: 0 of 5" gpurun_out/r5h_det.txt | head -10; tail -8 gpurun_out/r5h_soak.txt
