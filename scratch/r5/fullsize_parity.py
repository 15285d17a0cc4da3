"""Round 4: every problem of every bench batch against the serial twin (fp64 kernels), on the final build -- the two of
scratch/r3_fullsize_parity.py plus the horizons whose kernels changed this round (N = 40 / 60 / 80 tracking, IAC N = 40,
learning N = 40 / 60)."""
import sys, time, numpy as np, torch, importlib
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]; sys.path.insert(0, str(ROOT))
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


def tracking(kind, N, B):
    iac = kind == "iac"
    tr = pkg.workloads.synthetic_track("putnam" if iac else "barc")
    if iac:
        x, u = pkg.workloads.sample_initial_states("putnam", B, tr["L"], [-10.0, -0.314159], [5.0, 0.314159], seed=1)
        pc, pv, oc, ov = pkg.presets.iac_tracking_mpc(N), pkg.presets.iac_vehicle(), P.iac_tracking_mpc(N), P.iac_vehicle()
    else:
        x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], seed=0)
        pc, pv, oc, ov = pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), P.barc_tracking_mpc(N), P.barc_vehicle()
    sv = pkg.Solver(pc, pv, device=0)
    inp = sv.prepare(tr, x.T.copy(), 0.025); inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=dev)
    o = npd(sv.solve(inp)); t0 = time.time()
    tw = cbind.solve_batch(oc, ov, npd(inp))
    report("%s tracking N = %d (twin %.0f s)" % ("IAC" if iac else "BARC", N, time.time() - t0), o, tw)
    sv.close()


def learning(N, B):
    tr = pkg.workloads.synthetic_track("barc")
    cfg = dict(pkg.presets.barc_lmpc(N, 5)); laps = pkg.workloads.synthetic_laps(tr, 5)
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
    tw = cbind.solve_batch(P.barc_lmpc(N, 5), P.barc_vehicle(), npd(inp), ss_x.cpu().numpy(), ss_j.cpu().numpy())
    report("BARC learning N = %d, 160 points (twin %.0f s)" % (N, time.time() - t0), o, tw)
    sv.close()


tracking("barc", 20, 65536)
learning(20, 32768)
tracking("barc", 40, 4096)
tracking("barc", 60, 4096)
tracking("barc", 80, 4096)
tracking("iac", 40, 8192)
learning(40, 4096)
learning(60, 4096)
