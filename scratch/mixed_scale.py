"""Does cost scaling rescue the fp32 iteration on the BARC tracking problem?  (solution is invariant to the scale)"""
import sys, time, importlib
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
kk = o64["kkt"].cpu().numpy()
for c in (1.0, 10.0, 100.0, 1000.0, 1e4):
    cfg = dict(base)
    for k in ("q_contour", "q_heading", "q_vel", "q_vy", "q_vyaw", "q_boundary"):
        cfg[k] = base[k] * c
    cfg["R"] = [v * c for v in base["R"]]; cfg["R_d"] = [v * c for v in base["R_d"]]
    s = pkg.Solver(cfg, pkg.presets.barc_vehicle(), device=0)
    om = s.solve(inp, mixed=True)
    a, b = o64["status"].cpu().numpy(), om["status"].cpu().numpy()
    ok = (a == 0) & (b == 0)
    e = ((om["X_optm"] - o64["X_optm"]).abs().cpu().numpy() / SCALE_X[:, None, None]).max(axis=(0, 1))[ok]
    print(f"cost x{c:g}: solved {np.mean(b==0):.4f} (fp64 {np.mean(a==0):.4f}) iters {om['iters'].float().mean():.2f} err med {np.median(e):.2e} p90 {np.percentile(e,90):.2e} p99 {np.percentile(e,99):.2e} max {e.max():.2e}  status {np.bincount(b)}", flush=True)
