import sys, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
pkg = load_package()
N, B = 40, 8192
tr = pkg.workloads.synthetic_track("putnam")
x, u = pkg.workloads.sample_initial_states("putnam", B, tr["L"], [-10.0, -0.314159], [5.0, 0.314159], 1)
b = int(sys.argv[1])
for mi in list(range(12, 26)):
    cfg = pkg.presets.iac_tracking_mpc(N); cfg["max_iter"] = mi
    solver = pkg.Solver(cfg, pkg.presets.iac_vehicle(), 0)
    inp = solver.prepare(tr, x[b:b + 1].T.copy(), 0.025)
    inp["u_ic"] = torch.as_tensor(u[b:b + 1].T.copy(), dtype=torch.float64, device="cuda")
    o = solver.solve_f32(inp); o64 = solver.solve(inp)
    k = o["kkt"].cpu().numpy()[:, 0]; k64 = o64["kkt"].cpu().numpy()[:, 0]
    X = o["X_optm"].cpu().numpy()[:, :, 0]
    print("max_iter %2d: f32 status %d iters %2d step %.2e rd %.2e mu %.2e sigma %.2e | f64 it %d mu %.2e | X finite %s max|X| %.1e" % (mi, int(o["status"][0]), int(o["iters"][0]), k[0], k[1], k[2], k[3], int(o64["iters"][0]), k64[2], np.isfinite(X).all(), np.nanmax(np.abs(X[1:]))))
