#!/bin/bash
# round 5, GPU call 9: warm start -- its tests, the soak, and the cold headline next to it (the cold kernels must not have moved)
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
( time python -m pytest tests/test_gpu_warm.py tests/test_gpu_ss_idx.py -q -m gpu -s 2>&1 | grep -v "^$" | tail -30 ) > gpurun_out/r5i_pytest.log 2>&1
python scratch/r5/soak_warm.py > gpurun_out/r5i_soak_warm.txt 2>&1
python bench.py --no-others --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline', round(d['value']), d['ms_per_step'], d['ms_per_step_one_stream'], d['kernels_ms'])" > gpurun_out/r5i_bench.txt 2>&1
cat gpurun_out/r5i_pytest.log; grep -v amdgpu gpurun_out/r5i_soak_warm.txt; cat gpurun_out/r5i_bench.txt
