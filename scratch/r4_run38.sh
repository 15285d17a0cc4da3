#!/bin/bash
# round 4, GPU call 38: s_setprio also around the learning problem's terminal elimination (mixed learning kernels only)
AB=racing-lmpc-ros2_amd/lib/ab
for r in 1 2; do for v in tp0 tp1; do
LMPC_HIP_LIBRARY=$PWD/$AB/liblmpc_$v.so timeout 600 python scratch/r4_ab.py lmpc32kreg lmpc 2>&1 | grep -E '^\{' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    if r['prec'] == 'mixed': print('$v', r['case'], r['B'], r['prec'], r['qp_ms'], r['sha'], r.get('err_max'))
"
done; done
