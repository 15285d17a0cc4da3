"""Iteration-count / status histogram of the bench workload (GPU box)."""
import sys, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
pkg = load_package()
N, B = 20, 4096
solver = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), 0)
tr = pkg.workloads.synthetic_track("barc")
x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], 0)
inp = solver.prepare(tr, x.T.copy(), 0.025)
inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device="cuda")
out = solver.solve(inp)
torch.cuda.synchronize()
it = out["iters"].cpu().numpy(); st = out["status"].cpu().numpy(); k = out["kkt"].cpu().numpy()
print("status counts", np.bincount(st, minlength=3))
print("iters hist", np.bincount(it))
for s in (1, 2):
    m = st == s
    if m.any():
        print("status", s, "iters", np.bincount(it[m]), "rd", np.sort(k[1][m])[-5:], "mu", np.sort(k[2][m])[-5:])
m = (st == 0) & (it >= 18)
print("slow optimal:", m.sum(), "x0 sample", x[m][:5])
np.savez("gpurun_out/iter_hist.npz", it=it, st=st, x=x, u=u, kkt=k)
