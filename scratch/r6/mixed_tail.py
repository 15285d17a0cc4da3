"""Round 6: the problems of the spec learning batch (32768, no regression) whose mixed answer is > 3e-4 from the fp64 answer."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package
from parity import per_problem_err
import dense_cases as DC
pkg = load_package()
dev = "cuda"
tr = pkg.workloads.synthetic_track("barc")
laps = DC.spec_laps()
cfgd = pkg.presets.barc_lmpc(20, 5)
B = 32768
sv = pkg.Solver(cfgd, pkg.presets.barc_vehicle(), device=0)
sv.set_safe_set(laps, tr["L"])
x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], seed=0)
inp = sv.prepare(tr, x.T.copy(), 0.025)
inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=dev)
q = torch.as_tensor(DC.ss_query_point({k: inp[k].cpu().numpy() for k in ("X_ref", "x_ic")}, tr["L"]), device=dev).contiguous()
ss_x, ss_j, _ = sv.ss_query(q)
def solve(mixed):
    out = sv.alloc_outputs(B)
    out["convex_combi_optm"] = torch.zeros((160, B), dtype=torch.float64, device=dev)
    out["kkt"] = torch.zeros((4, B), dtype=torch.float64, device=dev)
    return {k: v.cpu().numpy() for k, v in sv.solve(inp, out, ss_x=ss_x, ss_j=ss_j, mixed=mixed).items() if hasattr(v, "cpu")}
o64, om = solve(False), solve(True)
K = ("X_optm", "U_optm", "dU_optm")
e, ed = per_problem_err({k: om[k] for k in K}, {k: o64[k] for k in K})
ok = (o64["status"] == 0) & (om["status"] == 0)
e[~ok] = 0
print("count > 1e-3:", (e > 1e-3).sum(), "> 3e-4:", (e > 3e-4).sum(), "> 1e-4:", (e > 1e-4).sum())
ssj = ss_j.cpu().numpy()
for b in np.argsort(e)[-12:][::-1]:
    l64, lm = o64["convex_combi_optm"][:, b], om["convex_combi_optm"][:, b]
    s64, sm = np.where(l64 > 1e-9)[0], np.where(lm > 1e-9)[0]
    print("b %5d e %.1e dU %.1e iters64 %2d itersM %2d kktM %s | support64 %s %s | supportM %s %s | x0 vx %.2f" % (
        b, e[b], ed[b], o64["iters"][b], om["iters"][b], om["kkt"][:, b].round(9).tolist(), s64.tolist(), l64[s64].round(4).tolist(), sm.tolist(), lm[sm].round(4).tolist(), x[b, 3]))
# the same with the fp32 pass's marks visible (polish = 1): were the bad ones verified by the fp32 KKT test?
cfg1 = dict(cfgd); cfg1["polish"] = 1
sv1 = pkg.Solver(cfg1, pkg.presets.barc_vehicle(), device=0)
sv1.set_safe_set(laps, tr["L"])
out = sv1.alloc_outputs(B); out["convex_combi_optm"] = torch.zeros((160, B), dtype=torch.float64, device=dev)
o1 = {k: v.cpu().numpy() for k, v in sv1.solve(inp, out, ss_x=ss_x, ss_j=ss_j, mixed=True).items() if hasattr(v, "cpu")}
print("marks (status 3) in the fp32 pass:", (o1["status"] == 3).sum(), "of", B)
bad = np.argsort(e)[-12:][::-1]
print("status of the worst in the fp32 pass:", o1["status"][bad].tolist())
