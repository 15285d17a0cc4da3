#!/bin/bash
# round 4, GPU call 29: linearisation kernel compiled for two waves per SIMD (spills) against one (1216 waves on 1024 slots at batch 4096)
mkdir -p gpurun_out
AB=racing-lmpc-ros2_amd/lib/ab
for v in lw1 lw2 lw1 lw2; do
  LMPC_HIP_LIBRARY=$PWD/$AB/liblmpc_$v.so timeout 600 python scratch/r4_ab.py trk20 trk20big lmpc32k iac 2>&1 | grep -E '^\{' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print('$v', r['case'], r['B'], r['prec'], 'lin', r['lin_ms'], 'qp', r['qp_ms'], r['sha'])
"
done
for v in lw1 lw2; do
  LMPC_HIP_LIBRARY=$PWD/$AB/liblmpc_$v.so python bench.py --steps 40 --no-others --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,3), round(d['ms_per_step'],4), 'one stream', round(d['ms_per_step_one_stream'],4), d['kernels_ms'])"
done
