"""Bitwise run-to-run reproducibility of the two-pass mixed solves (fp32 pass + collect + fp64 pass) and of the fp32 entry."""
import sys, numpy as np, torch, importlib
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
pkg = importlib.import_module("racing-lmpc-ros2_amd")
dev = torch.device("cuda:0")
def lmpc(B, N=20):
    tr = pkg.workloads.synthetic_track("barc"); cfg = dict(pkg.presets.barc_lmpc(N, 5)); laps = pkg.workloads.synthetic_laps(tr, 5)
    x, u = pkg.workloads.sample_states_near_laps(laps, B, tr["L"], seed=0)
    sv = pkg.Solver(cfg, pkg.presets.barc_vehicle(), device=0); sv.reserve(B)
    inp = sv.prepare(tr, x.T.copy(), 0.025); inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=dev)
    sv.set_safe_set(laps, tr["L"])
    s_last, s0, L = inp["X_ref"][0, -1], inp["x_ic"][0], tr["L"]
    kk = (s0 - s_last).abs() + L / 2
    q = torch.stack([s_last + (kk - torch.fmod(kk, L)) * torch.sign(s0 - s_last), inp["X_ref"][1, -1]]).contiguous()
    ss_x, ss_j, _ = sv.ss_query(q)
    def run():
        out = sv.alloc_outputs(B); out["convex_combi_optm"] = torch.zeros((int(cfg["num_ss_pts"]), B), dtype=torch.float64, device=dev)
        o = sv.solve(inp, out, mixed=True, ss_x=ss_x, ss_j=ss_j); torch.cuda.synchronize()
        return {k: v.clone() for k, v in o.items() if hasattr(v, "clone")}
    return run
def iac(B, kind):
    tr = pkg.workloads.synthetic_track("putnam"); cfg = dict(pkg.presets.iac_tracking_mpc(40)); veh = pkg.presets.iac_vehicle()
    x, u = pkg.workloads.sample_initial_states("putnam", B, tr["L"], [-10.0, -0.314159], [5.0, 0.314159], seed=1)
    sv = pkg.Solver(cfg, veh, device=0); sv.reserve(B)
    inp = sv.prepare(tr, x.T.copy(), 0.025); inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=dev)
    inp32 = {k: (v.float() if hasattr(v, "float") and v.dtype == torch.float64 else v) for k, v in inp.items()}
    def run():
        o = sv.solve(inp, sv.alloc_outputs(B), mixed=True) if kind == "mixed" else sv.solve_f32(inp32)
        torch.cuda.synchronize()
        return {k: v.clone() for k, v in o.items() if hasattr(v, "clone")}
    return run
for name, fn in (("learning 32768 mixed", lmpc(32768)), ("learning 4096 mixed", lmpc(4096)), ("iac 8192 mixed", iac(8192, "mixed")), ("iac 8192 f32", iac(8192, "f32"))):
    runs = [fn() for _ in range(6)]
    bad = sum(int(any((r[k] != runs[0][k]).any() for k in ("X_optm", "U_optm", "dU_optm", "status", "iters"))) for r in runs[1:])
    print(name, "runs differing bitwise from the first:", bad, "of 5; status", np.bincount(runs[0]["status"].cpu().numpy(), minlength=4).tolist())
