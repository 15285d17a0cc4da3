import sys, numpy as np, ctypes as C, os
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package
from oracle import cbind, params as P, qp as Q, scenario as S
pkg = load_package()
_real = C.CDLL(str(ROOT / "scratch/_exp_oracle.so")); cbind.lib = lambda: _real
veh, cfg = P.barc_vehicle(), P.barc_tracking_mpc(20)
tr = pkg.workloads.synthetic_track("barc")
u_lo, u_hi, _, _ = Q.effective_bounds(cfg, veh)
x, u = pkg.workloads.sample_initial_states("barc", 192, tr["L"], u_lo, u_hi, 5)
inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
b = int(sys.argv[1])
d = np.load("/tmp/acc_dense_192.npz")
for mi in range(4, 14):
    out = cbind.solve_batch(cfg, veh, inp, b0=b, b1=b + 1, max_iter=mi)
    ex = np.abs((out["X_optm"][:, :, b] - d["X"][:, :, b]) / P.SCALE_X[:, None]).max()
    print("max_iter", mi, "iters", out["iters"][b], "status", out["status"][b], "mu %.2e" % out["kkt"][2, b], "err vs dense %.2e" % ex)
