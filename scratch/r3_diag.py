"""Mixed learning solve, one pass (lmpc_config.polish = 1): which problems are accepted / marked, and how far the accepted
ones are from the fp64 answers; details of the worst accepted ones."""
import sys, os, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import importlib
pkg = importlib.import_module("racing-lmpc-ros2_amd")
import lmpc_scenario as LS
SX = np.array([2000, 10, 0.1, 80, 2, 2.0]); SU = np.array([10, 0.3])
dev = torch.device("cuda:0")
def err(a, b):
    ex = (torch.abs(a["X_optm"] - b["X_optm"]).cpu().numpy() / SX[:, None, None]).max(axis=(0, 1))
    eu = (torch.abs(a["U_optm"] - b["U_optm"]).cpu().numpy() / SU[:, None, None]).max(axis=(0, 1))
    return np.maximum(ex, eu)
def run(cfg, veh, inp, ss, B, S_pts, polish, mixed):
    c = dict(cfg); c["polish"] = polish
    sv = pkg.Solver(c, veh, device=0)
    out = sv.alloc_outputs(B); out["convex_combi_optm"] = torch.zeros((S_pts, B), dtype=torch.float64, device=dev)
    out["kkt"] = torch.zeros((4, B), dtype=torch.float64, device=dev)
    o = sv.solve(inp, out, ss_x=ss[0], ss_j=ss[1], mixed=mixed)
    torch.cuda.synchronize()
    r = {k: (v.clone() if hasattr(v, "clone") else v) for k, v in o.items()}
    sv.close()
    return r
# ---- the golden problems
g = dict(np.load(ROOT / "tests/golden/qp_barc_lmpc_n20.npz"))
cfg, veh = pkg.presets.barc_lmpc(20, 3), pkg.presets.barc_vehicle()
sv = pkg.Solver(cfg, veh, device=0); sv.set_safe_set(LS.load_laps(), LS.L_BARC_SS)
ss = sv.ss_query(g["query"])[:2]; sv.close()
B = g["x_ic"].shape[1]
ref = run(cfg, veh, g, ss, B, 96, 0, False)
for polish in (1, -1, 0):
    o = run(cfg, veh, g, ss, B, 96, polish, True)
    e = err(o, ref)
    print(f"golden mixed polish={polish}: status {o['status'].cpu().numpy()} iters {o['iters'].cpu().numpy()}")
    print("   err", np.array2string(e, precision=1), " kkt last_step", np.array2string(o['kkt'][0].cpu().numpy(), precision=1), "rd", np.array2string(o['kkt'][1].cpu().numpy(), precision=1), "mu", np.array2string(o['kkt'][2].cpu().numpy(), precision=1))
# ---- the large batch
tr = pkg.workloads.synthetic_track("barc"); laps = pkg.workloads.synthetic_laps(tr, 5)
cfg = pkg.presets.barc_lmpc(20, 5); B = 32768
x, u = pkg.workloads.sample_states_near_laps(laps, B, tr["L"], seed=0)
sv = pkg.Solver(cfg, veh, device=0); sv.set_safe_set(laps, tr["L"])
inp = sv.prepare(tr, x.T.copy(), 0.025); inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=dev)
s_last, s0, L = inp["X_ref"][0, -1], inp["x_ic"][0], tr["L"]
kk = (s0 - s_last).abs() + L / 2
q = torch.stack([s_last + (kk - torch.fmod(kk, L)) * torch.sign(s0 - s_last), inp["X_ref"][1, -1]]).contiguous()
ss = sv.ss_query(q)[:2]
def run2(polish, mixed):
    c = dict(cfg); c["polish"] = polish
    s2 = pkg.Solver(c, veh, device=0); s2.set_safe_set(laps, tr["L"])
    out = s2.alloc_outputs(B); out["convex_combi_optm"] = torch.zeros((160, B), dtype=torch.float64, device=dev); out["kkt"] = torch.zeros((4, B), dtype=torch.float64, device=dev)
    o = s2.solve(inp, out, ss_x=ss[0], ss_j=ss[1], mixed=mixed); torch.cuda.synchronize()
    r = {k: (v.clone() if hasattr(v, "clone") else v) for k, v in o.items()}; s2.close(); return r
ref = run2(0, False)
o = run2(1, True)
e = err(o, ref); st = o["status"].cpu().numpy(); ok = (ref["status"].cpu().numpy() == 0)
acc = ok & (st == 0)
print(f"batch {B}: one pass status {np.bincount(st, minlength=4)}; accepted max err {e[acc].max():.2e}; marked: {np.sum(st == 3)}, their fp32 err max {e[ok & (st == 3)].max():.2e}")
worst = np.argsort(-np.where(acc, e, 0))[:6]
ls = o["kkt"][0].cpu().numpy(); it = o["iters"].cpu().numpy(); itr = ref["iters"].cpu().numpy()
for b in worst: print(f"   b={b} err {e[b]:.2e} last_step {ls[b]:.2e} iters {it[b]} (fp64 {itr[b]}) mu {o['kkt'][2][b].item():.1e} rd {o['kkt'][1][b].item():.1e}")
for lo, hi in ((0, 1e-7), (1e-7, 1e-6), (1e-6, 1e-5), (1e-5, 1e-4), (1e-4, 1e9)):
    sel = acc & (ls >= lo) & (ls < hi)
    if sel.any(): print(f"     last step in [{lo:g},{hi:g}): {sel.sum():6d} problems, max err {e[sel].max():.2e}")
np.savez("gpurun_out/r3_diag_worst.npz", idx=worst, **{k: inp[k][..., torch.as_tensor(worst, device=dev)].cpu().numpy() for k in inp if hasattr(inp[k], "shape") and inp[k].ndim >= 1 and inp[k].shape[-1] == B}, ss_x=ss[0][..., torch.as_tensor(worst, device=dev)].cpu().numpy(), ss_j=ss[1][..., torch.as_tensor(worst, device=dev)].cpu().numpy())
