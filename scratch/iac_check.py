import sys, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package
from oracle import cbind, params as P, qp as Q, scenario as S
pkg = load_package()
d = np.load("gpurun_out/iac_status.npz")
veh, cfg = P.iac_vehicle(), P.iac_tracking_mpc(40)
tr = pkg.workloads.synthetic_track("putnam")
idx = np.where(d["st"] == 2)[0][:4]
x, u = d["x"][idx], d["u"][idx]
inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
tw = cbind.solve_batch(cfg, veh, inp)
print("twin status", tw["status"], tw["iters"])
for b in range(len(idx)):
    qp = Q.build_qp(cfg, veh, S.problem(inp, b))
    y, info = Q.solve_dense(qp)
    print(idx[b], "dense status", info["status"], "iters", info.get("iters"), {k: ("%.2e" % v) for k, v in info.items() if isinstance(v, float)})
    Xr = inp["X_ref"][:, :, b]
    print("   X_ref vx range %.1f..%.1f  vy %.2f..%.2f  w %.2f..%.2f  vel_ref %.1f..%.1f curv max %.4f" % (Xr[3].min(), Xr[3].max(), Xr[4].min(), Xr[4].max(), Xr[5].min(), Xr[5].max(), inp["vel_ref"][:, b].min(), inp["vel_ref"][:, b].max(), np.abs(inp["curvatures"][:, b]).max()))
