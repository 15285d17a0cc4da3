import sys, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
pkg = load_package()
N, B = 40, 8192
solver = pkg.Solver(pkg.presets.iac_tracking_mpc(N), pkg.presets.iac_vehicle(), 0)
tr = pkg.workloads.synthetic_track("putnam")
x, u = pkg.workloads.sample_initial_states("putnam", B, tr["L"], [-10.0, -0.314159], [5.0, 0.314159], seed=1)
inp = solver.prepare(tr, x.T.copy(), 0.025)
inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device="cuda")
out = solver.solve(inp)
st = out["status"].cpu().numpy(); it = out["iters"].cpu().numpy(); k = out["kkt"].cpu().numpy()
print("status", np.bincount(st, minlength=3)); print("iters", np.bincount(it))
for s in (1, 2):
    m = st == s
    if m.any():
        print("status", s, "iters hist", np.bincount(it[m]), "rd pct", np.percentile(k[1][m], [10, 50, 90]), "mu pct", np.percentile(k[2][m], [10, 50, 90]))
        idx = np.where(m)[0][:6]
        print(" examples x0:", x[idx].round(3), "u0", u[idx].round(4))
np.savez("gpurun_out/iac_status.npz", st=st, it=it, x=x, u=u, kkt=k)
