#!/bin/bash
# round 5, GPU call 14: per-kernel trace of the closed loop (cold / warm), the long horizon in fp32 / mixed, the suite once more
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
bash scratch/r5/loop_prof.sh > gpurun_out/loop_prof.log 2>&1
for p in mixed f32; do
  python bench.py --horizon 60 --precision $p --no-others --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_bench_tracking_n60_$p.json
done
( time python -m pytest tests -q -m gpu -x ) > gpurun_out/r05_gpu_suite.txt 2>&1
cat gpurun_out/r05_closed_loop_rocprof_summary.md | cut -c1-200
tail -5 gpurun_out/r05_gpu_suite.txt
for p in mixed f32; do python - $p <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r05_bench_tracking_n60_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]), "ms", round(d["ms_per_step"],3), "one stream", round(d["ms_per_step_one_stream"],3), d.get("kernels_ms") or d.get("kernels"), "solved", d.get("solved_fraction"), "iters", d.get("mean_iters"))
PY
done
