#!/bin/bash
# round 4, GPU call 35: s_setprio around the serial stage chains (factor, sweeps) of the fat kernels: 0 (shipped), 1, 3
AB=racing-lmpc-ros2_amd/lib/ab
run() { v=$1; shift; LMPC_HIP_LIBRARY=$PWD/$AB/liblmpc_$v.so timeout 600 python scratch/r4_ab.py "$@" 2>&1 | grep -E '^\{' ; }
{ for rep in 1 2; do for v in pr0 pr1 pr3; do run $v trk20 trk20big lmpc32kreg iac; done; done; } > gpurun_out/r4ad_ab.jsonl
python - <<'PY'
import json, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r4ad_ab.jsonl"):
    r = json.loads(l)
    if "error" in r: print(r); continue
    d[(r["case"], r["B"], r["prec"], r["lib"])].append((r["qp_ms"], r["sha"]))
libs = ["pr0", "pr1", "pr3"]
print("| case | " + " | ".join(libs) + " |")
for k in sorted(set(k[:3] for k in d)):
    ref = d[k + ("liblmpc_pr0.so",)][0][1]
    print("| %s %d %s | " % k + " | ".join("/".join("%.3f" % m for m, _ in d[k + ("liblmpc_%s.so" % l,)]) + ("=" if all(s == ref for _, s in d[k + ("liblmpc_%s.so" % l,)]) else "!") for l in libs) + " |")
PY
for v in pr0 pr1 pr3; do LMPC_HIP_LIBRARY=$PWD/$AB/liblmpc_$v.so python bench.py --steps 60 --no-others --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v pipelined', round(d['value']/1e6,3), 'one stream', round(d['ms_per_step_one_stream'],4))"; done
