import sys, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package
from oracle import cbind, params as P, qp as Q, scenario as S
pkg = load_package()
veh, cfg = P.barc_vehicle(), P.barc_tracking_mpc(20)
tr = pkg.workloads.synthetic_track("barc")
u_lo, u_hi, v_lo, v_hi = Q.effective_bounds(cfg, veh)
x, u = pkg.workloads.sample_initial_states("barc", 192, tr["L"], u_lo, u_hi, 5)
inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
d = np.load("/tmp/acc_dense_192.npz")
X, U, D = d["X"], d["U"], d["D"]
print("u bounds", u_lo, u_hi, "v bounds", v_lo, v_hi, "x min", cfg.x_min, "x max", cfg.x_max)
tot = np.zeros(11)
for b in list(range(0, 192)):
    Xb, Ub, Db = X[:, :, b], U[:, :, b], D[:, :, b]
    cnt = np.zeros(11, int)
    for k in range(6):
        cnt[k] = ((np.abs(Xb[k, 1:-1] - cfg.x_max[k]) < 1e-7) | (np.abs(Xb[k, 1:-1] - cfg.x_min[k]) < 1e-7)).sum()
    for k in range(2):
        cnt[6 + k] = ((np.abs(Ub[k] - u_hi[k]) < 1e-7) | (np.abs(Ub[k] - u_lo[k]) < 1e-7)).sum()
        cnt[8 + k] = ((np.abs(Db[k] - v_hi[k]) < 1e-7) | (np.abs(Db[k] - v_lo[k]) < 1e-7)).sum()
    bl, br = inp["bound_left"][:, b], inp["bound_right"][:, b]
    cnt[10] = ((Xb[1] > bl - cfg.margin - 1e-6) | (Xb[1] < br + cfg.margin + 1e-6)).sum() if hasattr(cfg, "margin") else -1
    tot += cnt
    if b in (131, 54, 119, 154, 144): print(b, "active rows per kind [s ey epsi vx vy w | u0 u1 | v0 v1 | bnd]", cnt)
print("total over 192 problems", tot)
