"""Bitwise reproducibility soak at bench sizes: tracking N = 20 / 40 and the learning problem (160 points), many repetitions."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from __graft_entry__ import load_package
pkg = load_package()
import test_gpu_mixed_lmpc as T
def soak(name, solve, reps):
    ref = None; bad = 0
    for k in range(reps):
        o = solve()
        cur = (o["X_optm"].clone(), o["U_optm"].clone(), o["iters"].clone(), o["status"].clone())
        if ref is None: ref = cur
        else: bad += int(not all(torch.equal(a, b) for a, b in zip(cur, ref)))
    print(name, "reps", reps, "runs differing from the first:", bad)
tr = pkg.workloads.synthetic_track("barc")
import os
REPS20 = int(os.environ.get('REPS20', '40'))
for N, reps in ((20, REPS20), (40, 12)):
    sv = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), device=0)
    x, u = pkg.workloads.sample_initial_states("barc", 4096, tr["L"], [-0.01, -0.314], [0.01, 0.314], seed=0)
    inp = sv.prepare(tr, x.T.copy(), 0.025); inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device="cuda")
    soak("tracking N=%d B=4096" % N, lambda: sv.solve(inp), reps)
sv, tr2, laps, inp, ss_x, ss_j = T._s160(pkg, 4096)
soak("learning S=160 B=4096 fp64", lambda: sv.solve(inp, ss_x=ss_x, ss_j=ss_j), 20)
soak("learning S=160 B=4096 mixed", lambda: sv.solve(inp, ss_x=ss_x, ss_j=ss_j, mixed=True), 20)
