#!/bin/bash
# round 4, GPU call 28: regression kernel with the group's scalar loads issued up front (one wait per four samples)
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
python -m pytest tests/test_gpu_regression.py tests/test_gpu_mixed_lmpc.py -x -q -m gpu 2>&1 | tail -2
O=gpurun_out
run() { name=$1; shift; python bench.py "$@" 2>$O/r04_bench_$name.err | tail -1 > $O/r04_bench_$name.json; }
run lmpc_b32768_mixed_regression --workload lmpc --batch 32768 --precision mixed --regression --steps 10 --no-others --no-cpu-baseline
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_lmpc_b32768_mixed_regression.json"))
print(d["value"], d["ms_per_step"], d["ms_per_step_one_stream"], d["kernels_ms"])
PY
bash scratch/prof.sh lmpcmix --workload lmpc --batch 32768 --precision mixed --regression > gpurun_out/prof_lmpcmix.log 2>&1
grep -E "regress|solve_kernel" gpurun_out/prof_lmpcmix/summary.md | head -3
