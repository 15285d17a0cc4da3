"""fp64 second pass on the WHOLE batch (LMPC_DEBUG_CLEANUP_ALL=1) against the direct fp64 kernel: must be the same bits."""
import sys, os, numpy as np, torch, importlib
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
pkg = importlib.import_module("racing-lmpc-ros2_amd")
dev = torch.device("cuda:0")
def run(N, B, mixed, n_laps=5):
    tr = pkg.workloads.synthetic_track("barc")
    if n_laps == 0:
        return run_trk(N, B, mixed)
    cfg = dict(pkg.presets.barc_lmpc(N, n_laps)); laps = pkg.workloads.synthetic_laps(tr, n_laps)
    x, u = pkg.workloads.sample_states_near_laps(laps, B, tr["L"], seed=0)
    sv = pkg.Solver(cfg, pkg.presets.barc_vehicle(), device=0); sv.reserve(B)
    inp = sv.prepare(tr, x.T.copy(), 0.025); inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=dev)
    sv.set_safe_set(laps, tr["L"])
    s_last, s0, L = inp["X_ref"][0, -1], inp["x_ic"][0], tr["L"]
    kk = (s0 - s_last).abs() + L / 2
    q = torch.stack([s_last + (kk - torch.fmod(kk, L)) * torch.sign(s0 - s_last), inp["X_ref"][1, -1]]).contiguous()
    ss_x, ss_j, _ = sv.ss_query(q)
    out = sv.alloc_outputs(B); out["convex_combi_optm"] = torch.zeros((int(cfg["num_ss_pts"]), B), dtype=torch.float64, device=dev)
    o = sv.solve(inp, out, mixed=mixed, ss_x=ss_x, ss_j=ss_j); torch.cuda.synchronize()
    return {k: v.cpu().numpy().copy() for k, v in o.items() if hasattr(v, "cpu")}
def run_trk(N, B, mixed):
    tr = pkg.workloads.synthetic_track("barc")
    cfg = dict(pkg.presets.barc_tracking_mpc(N)); veh = pkg.presets.barc_vehicle()
    x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], seed=0)
    sv = pkg.Solver(cfg, veh, device=0); sv.reserve(B)
    inp = sv.prepare(tr, x.T.copy(), 0.025); inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=dev)
    o = sv.solve(inp, sv.alloc_outputs(B), mixed=mixed); torch.cuda.synchronize()
    return {k: v.cpu().numpy().copy() for k, v in o.items() if hasattr(v, "cpu")}
for N, B, nl in ((40, 2048, 5), (40, 2048, 3), (20, 2048, 5), (20, 2048, 3), (20, 2048, 0), (40, 2048, 0), (60, 2048, 0), (80, 2048, 0)):
    a = run(N, B, False, nl); b = run(N, B, True, nl)
    print(N, "laps", nl, "direct", np.bincount(a["status"], minlength=4), "cleanup-all", np.bincount(b["status"], minlength=4), "iters equal", (a["iters"] == b["iters"]).mean(),
          "max |dX|", np.abs(a["X_optm"] - b["X_optm"]).max(), "bad", np.nonzero(a["status"] != b["status"])[0][:10])
