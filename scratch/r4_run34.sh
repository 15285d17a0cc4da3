#!/bin/bash
# round 4, GPU call 34: the mixed learning kernels at N = 40 / 60 again (-DLMPC_MIXED_LONG_LEARNING), after the row-phase changes and
# with the tightened acceptance test
AB=racing-lmpc-ros2_amd/lib/ab
python - <<'PY'
import re
p = "scratch/r4_ab.py"
s = open(p).read()
s = s.replace('"lmpc40": lambda: case("lmpc", 40, 4096, ("f64",)),', '"lmpc40": lambda: case("lmpc", 40, 4096, ("f64", "mixed")),')
s = s.replace('"lmpc60": lambda: case("lmpc", 60, 4096, ("f64",)),', '"lmpc60": lambda: case("lmpc", 60, 4096, ("f64", "mixed")),')
open("/tmp/r4_ab_mll.py", "w").write(s.replace('ROOT = Path(__file__).resolve().parents[1]', 'ROOT = Path("%s")' % __import__("os").getcwd()))
PY
LMPC_HIP_LIBRARY=$PWD/$AB/liblmpc_mll.so timeout 900 python /tmp/r4_ab_mll.py lmpc40 lmpc60 2>&1 | grep -E '^\{' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print({k: r.get(k) for k in ('case','B','prec','qp_ms','status','iters_mean','err_med','err_999','err_max','n_gt_1e3','lost_idx','error')})
"
