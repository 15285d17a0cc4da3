#!/bin/bash
# round 5, GPU call 13: the suite on the final host code (warm-rounds default), the learning lines with live counters, the soak on the default rounds
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
( time python -m pytest tests -q -m gpu --durations=5 ) > gpurun_out/r05_gpu_suite.txt 2>&1
python scratch/r5/soak_warm.py 2>&1 | grep -v amdgpu > gpurun_out/r05_closed_loop_soak.txt
( time python bench.py --workload lmpc --no-others --no-cpu-baseline ) 2> gpurun_out/lmpc.err | tail -1 > gpurun_out/r05_bench_lmpc.json
python bench.py --workload lmpc --laps 3 --no-others --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_bench_lmpc_laps3.json
python bench.py --workload lmpc --batch 32768 --precision mixed --regression --steps 10 --no-others --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_bench_lmpc_b32768_mixed_regression.json
tail -12 gpurun_out/r05_gpu_suite.txt; cat gpurun_out/r05_closed_loop_soak.txt; tail -4 gpurun_out/lmpc.err
for f in r05_bench_lmpc.json r05_bench_lmpc_laps3.json r05_bench_lmpc_b32768_mixed_regression.json; do python - $f <<'PY'
import json,sys
d=json.loads(open("gpurun_out/"+sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[1], round(d["value"]), "ms", round(d["ms_per_step"],3), d.get("kernels_ms") or d.get("kernels"), "traffic", r.get("traffic"), (r.get("traffic_source") or "")[:40], "x algo", r.get("traffic_over_algorithmic"), "algo B", r.get("algorithmic_bytes_per_solve"), str(r.get("engines"))[:200])
PY
done
