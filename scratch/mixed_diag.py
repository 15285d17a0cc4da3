import sys, importlib
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
pkg = importlib.import_module("racing-lmpc-ros2_amd")
SCALE_X = np.array([2000.0, 10.0, 0.1, 80.0, 2.0, 2.0])
dev = torch.device("cuda", 0)
B = 4096
tr = pkg.workloads.synthetic_track("barc")
base = pkg.presets.barc_tracking_mpc(20)
s64 = pkg.Solver(base, pkg.presets.barc_vehicle(), device=0)
x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.015, -0.314159], [0.015, 0.314159], seed=0)
inp = s64.prepare(tr, x.T.copy(), 0.025); inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=dev)
o64 = s64.solve(inp)
om = s64.solve(inp, mixed=True)
st = om["status"].cpu().numpy(); it = om["iters"].cpu().numpy(); kk = om["kkt"].cpu().numpy(); k64 = o64["kkt"].cpu().numpy(); it64 = o64["iters"].cpu().numpy()
e = ((om["X_optm"] - o64["X_optm"]).abs().cpu().numpy() / SCALE_X[:, None, None]).max(axis=(0, 1))
for code in (1, 2):
    idx = np.where(st == code)[0][:12]
    print("status", code, "count", (st == code).sum())
    for i in idx:
        print(f"  b={i} it={it[i]} (fp64 it={it64[i]} sigma={k64[3,i]:.3e}) step={kk[0,i]:.2e} rd={kk[1,i]:.2e} mu={kk[2,i]:.2e} sigma={kk[3,i]:.3e} x0={x[i].round(3)}")
print("iters hist of infeasible:", np.bincount(it[st == 2]))
bad = np.where((st == 0) & (e > 1e-2))[0][:10]
for i in bad:
    print(f"  inaccurate b={i} err={e[i]:.2e} it={it[i]} (fp64 it={it64[i]} sigma={k64[3,i]:.3e}) step={kk[0,i]:.2e} rd={kk[1,i]:.2e} mu={kk[2,i]:.2e} sigma={kk[3,i]:.3e}")
