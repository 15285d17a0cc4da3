import sys, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
pkg = load_package()
solver = pkg.Solver(pkg.presets.barc_tracking_mpc(20), pkg.presets.barc_vehicle(), device=0)
tr = pkg.workloads.synthetic_track("barc")
rng = np.random.default_rng(2); B = 512
x0 = np.stack([rng.uniform(0, tr["L"], B), rng.uniform(-0.1, 0.1, B), rng.normal(0, 0.03, B), rng.uniform(2.0, 2.5, B), np.zeros(B), np.zeros(B)])
trk = solver.device_track(tr)
x = torch.as_tensor(x0, device="cuda"); u_prev = torch.zeros((2, B), dtype=torch.float64, device="cuda")
inp = solver.prepare(trk, x, 0.025, speed_scale=0.9); out = solver.alloc_outputs(B)
for k in range(1100):
    inp["x_ic"] = x; inp["u_ic"] = u_prev
    solver.solve(inp, out)
    ok = out["status"] == 0
    u_apply = torch.where(ok[None, :], out["U_optm"][:, 0, :], inp["U_ref"][:, 0, :]).contiguous()
    nanin = [kk for kk in ("X_ref","U_ref","bound_left","bound_right","curvatures","vel_ref") if torch.isnan(inp[kk]).any()]
    nanout = [kk for kk in ("X_optm","U_optm") if torch.isnan(out[kk][..., ok]).any()]
    if k % 25 == 0 or nanin or nanout or (~ok).sum() > 0:
        st = out["status"].cpu().numpy()
        print(k, "status", np.bincount(st, minlength=3), "iters max", int(out["iters"].max()), "vx %.2f..%.2f" % (float(x[3].min()), float(x[3].max())), "|ey| max %.3f" % float(x[1].abs().max()), "nan in", nanin, "nan out", nanout, "u nan", bool(torch.isnan(u_apply).any()))
        if nanin or nanout: 
            bad = torch.nonzero(torch.isnan(inp["X_ref"]).any(0).any(0)).flatten()[:5]; print("bad cars", bad.tolist(), out["status"][bad].tolist())
            break
    solver.plant_step(trk, x, u_apply, 0.0125, 2)
    u_prev = u_apply
    inp = solver.shift(trk, inp, out, 0.025, speed_scale=0.9)
