"""Every problem of the largest bench batches against the serial twin: tracking 65536, learning 32768 (fp64 kernel)."""
import sys, time, numpy as np, torch, importlib
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from oracle import cbind, params as P
pkg = importlib.import_module("racing-lmpc-ros2_amd")
dev = torch.device("cuda:0")
SX, SU = P.SCALE_X[:, None, None], P.SCALE_U[:, None, None]
def report(name, o, tw):
    ok = (o["status"] == 0) & (tw["status"] == 0)
    ex = np.abs((o["X_optm"] - tw["X_optm"]) / SX).max(axis=(0, 1))[ok]; eu = np.abs((o["U_optm"] - tw["U_optm"]) / SU).max(axis=(0, 1))[ok]
    ed = np.abs((o["dU_optm"] - tw["dU_optm"]) / SU).max(axis=(0, 1))[ok]
    di = np.abs(o["iters"][ok] - tw["iters"][ok])
    print(f"{name}: {len(ok)} problems, kernel status {np.bincount(o['status'], minlength=3).tolist()} twin {np.bincount(tw['status'], minlength=3).tolist()} "
          f"status equal {float((o['status'] == tw['status']).mean()):.6f}; max scaled |dX| {ex.max():.1e} |dU| {eu.max():.1e} |d(dU)| {ed.max():.1e}; "
          f"iterations equal {float((di == 0).mean()):.4f}, within one {float((di <= 1).mean()):.4f}, max difference {int(di.max())}", flush=True)
def npd(d): return {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in d.items()}
# tracking
B = 65536
tr = pkg.workloads.synthetic_track("barc")
x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], seed=0)
sv = pkg.Solver(pkg.presets.barc_tracking_mpc(20), pkg.presets.barc_vehicle(), device=0)
inp = sv.prepare(tr, x.T.copy(), 0.025); inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=dev)
o = npd(sv.solve(inp)); t0 = time.time()
tw = cbind.solve_batch(P.barc_tracking_mpc(20), P.barc_vehicle(), npd(inp)); print("twin: %.1f s" % (time.time() - t0))
report("BARC tracking N = 20", o, tw)
# learning
B = 32768
cfg = dict(pkg.presets.barc_lmpc(20, 5)); laps = pkg.workloads.synthetic_laps(tr, 5)
x, u = pkg.workloads.sample_states_near_laps(laps, B, tr["L"], seed=0)
sv = pkg.Solver(cfg, pkg.presets.barc_vehicle(), device=0); sv.reserve(B)
inp = sv.prepare(tr, x.T.copy(), 0.025); inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=dev)
sv.set_safe_set(laps, tr["L"])
s_last, s0, L = inp["X_ref"][0, -1], inp["x_ic"][0], tr["L"]
kk = (s0 - s_last).abs() + L / 2
q = torch.stack([s_last + (kk - torch.fmod(kk, L)) * torch.sign(s0 - s_last), inp["X_ref"][1, -1]]).contiguous()
ss_x, ss_j, _ = sv.ss_query(q)
out = sv.alloc_outputs(B); out["convex_combi_optm"] = torch.zeros((160, B), dtype=torch.float64, device=dev)
o = npd(sv.solve(inp, out, ss_x=ss_x, ss_j=ss_j)); t0 = time.time()
tw = cbind.solve_batch(P.barc_lmpc(20, 5), P.barc_vehicle(), npd(inp), ss_x.cpu().numpy(), ss_j.cpu().numpy()); print("twin: %.1f s" % (time.time() - t0))
report("BARC learning N = 20, 160 points", o, tw)
