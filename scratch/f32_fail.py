import sys, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
pkg = load_package()
N, B = 40, 8192
solver = pkg.Solver(pkg.presets.iac_tracking_mpc(N), pkg.presets.iac_vehicle(), 0)
tr = pkg.workloads.synthetic_track("putnam")
x, u = pkg.workloads.sample_initial_states("putnam", B, tr["L"], [-10.0, -0.314159], [5.0, 0.314159], 1)
inp = solver.prepare(tr, x.T.copy(), 0.025)
inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device="cuda")
o32 = solver.solve_f32(inp); o64 = solver.solve(inp)
st = o32["status"].cpu().numpy(); it = o32["iters"].cpu().numpy(); k = o32["kkt"].cpu().numpy()
it64 = o64["iters"].cpu().numpy()
print("f32 iters hist", np.bincount(it))
for b in np.where(st != 0)[0]:
    print("b", b, "status", st[b], "iters", it[b], "rd %.2e mu %.2e" % (k[1, b], k[2, b]), "f64 iters", it64[b], "x0", x[b].round(2))
slow = np.argsort(it)[-5:]
print("slowest f32:", [(int(b), int(it[b]), int(it64[b]), "%.1e" % k[1, b], "%.1e" % k[2, b]) for b in slow])
