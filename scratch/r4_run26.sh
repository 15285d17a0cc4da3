#!/bin/bash
# round 4, GPU call 26: LMPC_OPAQUE_SITES variants on the short-horizon tracking kernels (two passes over the builds: timings between
# processes of one call differ by about 1 %)
mkdir -p gpurun_out
AB=racing-lmpc-ros2_amd/lib/ab
run() { v=$1; shift; LMPC_HIP_LIBRARY=$PWD/$AB/liblmpc_$v.so timeout 600 python scratch/r4_ab.py "$@" 2>&1 | grep -E '^\{' ; }
{
for rep in 1 2; do for v in s0 s15 s13 s5 s8 s7; do run $v trk20 trk20big trk40 iac; done; done
} > gpurun_out/r4y_ab.jsonl
python - <<'PY'
import json, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r4y_ab.jsonl"):
    r = json.loads(l)
    if "error" in r: print(r); continue
    d[(r["case"], r["B"], r["prec"], r["lib"])].append((r["qp_ms"], r["sha"]))
keys = sorted(set(k[:3] for k in d))
libs = ["s0", "s15", "s13", "s5", "s8", "s7"]
print("| case | " + " | ".join(libs) + " |")
for k in keys:
    ref = d[k + ("liblmpc_s0.so",)][0][1]
    print("| %s %d %s | " % k + " | ".join("/".join("%.3f" % m for m, _ in d[k + ("liblmpc_%s.so" % l,)]) + ("=" if all(s == ref for _, s in d[k + ("liblmpc_%s.so" % l,)]) else "!") for l in libs) + " |")
PY
