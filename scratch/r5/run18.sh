#!/bin/bash
# round 5, GPU call 20: the default bench line of the final build; one car through the node core (warm keys in use); the C ABI from C++
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
( time python bench.py ) > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
L=racing-lmpc-ros2_amd/lib
T=tests/golden/barc_track/15_barc_optm.txt
{
for N in 20 40 60; do for mode in CONTINUOUS STEP; do $L/test_node_core $T $N 2.1 $mode 2>&1 | grep -i "laps\|PASS\|FAIL" | tail -2; done; done
$L/bench_cabi $T 4096 20 2>&1 | tail -2
$L/bench_cabi $T 4096 20 --gpus 2 --same-device 2>&1 | tail -3
} > gpurun_out/r05_node_core_and_cabi.txt 2>&1
tail -3 gpurun_out/r05_bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_bench_default.json").read().strip().splitlines()[-1])
print(round(d["value"]), d["ms_per_step"], d["ms_per_step_one_stream"], d.get("kernels_ms"), d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"])
print(json.dumps(d.get("closed_loop")))
PY
cat gpurun_out/r05_node_core_and_cabi.txt
