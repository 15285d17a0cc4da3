#!/bin/bash
# every bench line of the round (run on the GPU box from the repo root); lines land in gpurun_out/r06_bench_*.json
cd $GRAFT_REPO_ROOT
O=gpurun_out
run() { name=$1; shift; python bench.py "$@" 2>$O/r06_bench_$name.err | tail -1 > $O/r06_bench_$name.json; }
run tracking
run tracking_s1 --streams 1 --no-others --no-cpu-baseline
run tracking_b65536 --batch 65536 --steps 20 --no-others --no-cpu-baseline
run tracking_n40 --horizon 40 --no-others --no-cpu-baseline --steps 20
run tracking_n60 --horizon 60 --no-others --no-cpu-baseline --steps 10
run tracking_n80 --horizon 80 --no-others --no-cpu-baseline --steps 10
run lmpc --workload lmpc --no-others --no-cpu-baseline
run lmpc_b32768 --workload lmpc --batch 32768 --steps 10 --no-others --no-cpu-baseline
run lmpc_b32768_mixed --workload lmpc --batch 32768 --precision mixed --steps 10 --no-others --no-cpu-baseline
run lmpc_b32768_mixed_regression --workload lmpc --batch 32768 --precision mixed --regression --steps 10 --no-others --no-cpu-baseline
run iac_n40 --workload iac --horizon 40 --batch 8192 --steps 20 --no-others --no-cpu-baseline
run iac_n40_f32 --workload iac --horizon 40 --batch 8192 --precision f32 --steps 20 --no-others --no-cpu-baseline
run iac_n40_mixed --workload iac --horizon 40 --batch 8192 --precision mixed --steps 20 --no-others --no-cpu-baseline
for f in $O/r06_bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print(sys.argv[1].split("r06_bench_")[1], d["value"], d["unit"], "ms/step", d["ms_per_step"], "one stream", d.get("ms_per_step_one_stream"), "kernels", d.get("kernels_ms"), "frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
