#!/bin/bash
# round-2 measurement set (run on the GPU box from the repo root)
mkdir -p gpurun_out/r02
B="python bench.py --no-cpu-baseline --no-pmc"
python bench.py > gpurun_out/r02/bench_tracking.json 2> gpurun_out/r02/bench_tracking.err
$B --streams 1 > gpurun_out/r02/bench_tracking_s1.json 2>/dev/null
$B --batch 65536 --steps 10 --warmup 2 > gpurun_out/r02/bench_tracking_b65536.json 2>/dev/null
$B --workload lmpc > gpurun_out/r02/bench_lmpc.json 2>/dev/null
$B --workload lmpc --batch 32768 --steps 10 --warmup 2 > gpurun_out/r02/bench_lmpc_b32768.json 2>/dev/null
$B --workload lmpc --batch 32768 --steps 10 --warmup 2 --precision mixed > gpurun_out/r02/bench_lmpc_b32768_mixed.json 2>/dev/null
$B --workload lmpc --batch 32768 --steps 10 --warmup 2 --precision mixed --regression > gpurun_out/r02/bench_lmpc_b32768_mixed_regression.json 2>/dev/null
$B --workload lmpc --horizon 40 --steps 10 --warmup 2 > gpurun_out/r02/bench_lmpc_n40.json 2>/dev/null
$B --workload iac --horizon 40 --batch 8192 > gpurun_out/r02/bench_iac_n40.json 2>/dev/null
$B --workload iac --horizon 40 --batch 8192 --precision mixed > gpurun_out/r02/bench_iac_n40_mixed.json 2>/dev/null
$B --workload iac --horizon 40 --batch 8192 --precision f32 > gpurun_out/r02/bench_iac_n40_f32.json 2>/dev/null
$B --horizon 40 --steps 10 --warmup 2 > gpurun_out/r02/bench_tracking_n40.json 2>/dev/null
$B --horizon 60 --steps 10 --warmup 2 > gpurun_out/r02/bench_tracking_n60.json 2>/dev/null
bash scratch/prof.sh r02_tracking > /dev/null 2>&1
bash scratch/prof.sh r02_lmpc --workload lmpc > /dev/null 2>&1
for f in gpurun_out/r02/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], d.get("roofline",{}).get("kernel_ms"), d.get("solved_fraction"), d.get("mean_iterations"))
except Exception as e: print("ERR", e)
PY
done
