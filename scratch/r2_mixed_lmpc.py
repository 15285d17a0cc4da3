"""fp32 interior-point iteration between fp64 arrays on the LEARNING problem (BASELINE configs[4]) against the golden
vectors (S = 96) and the fp64 kernel on a fresh batch (S = 160)."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from __graft_entry__ import load_package
pkg = load_package()
from oracle import params as P
from parity import per_problem_err
import lmpc_scenario as LS
g = dict(np.load("/root/repo/tests/golden/qp_barc_lmpc_n20.npz"))
solver = pkg.Solver(pkg.presets.barc_lmpc(20, 3), pkg.presets.barc_vehicle(), device=0)
solver.set_safe_set(LS.load_laps(), LS.L_BARC_SS)
ss_x, ss_j, nf = solver.ss_query(g["query"])
def run(sv, inp, ssx, ssj, S, B, mixed):
    out = sv.alloc_outputs(B)
    out["convex_combi_optm"] = torch.zeros((S, B), dtype=torch.float64, device="cuda")
    return {k: v.cpu().numpy() for k, v in sv.solve(inp, out, ss_x=ssx, ss_j=ssj, mixed=mixed).items() if hasattr(v, "cpu")}
o = run(solver, g, ss_x, ss_j, 96, 16, True)
e, ed = per_problem_err(o, g)
print("golden S=96 mixed: status", o["status"], "iters", o["iters"], "err", " ".join(f"{x:.0e}" for x in e))
tr = pkg.workloads.synthetic_track("barc")
laps = pkg.workloads.synthetic_laps(tr, 5)
sv5 = pkg.Solver(pkg.presets.barc_lmpc(20, 5), pkg.presets.barc_vehicle(), device=0)
sv5.set_safe_set(laps, tr["L"])
B = 4096
x, u = pkg.workloads.sample_states_near_laps(laps, B, tr["L"], seed=0)
inp = sv5.prepare(tr, x.T.copy(), 0.025); inp["u_ic"] = torch.as_tensor(u.T.copy(), device="cuda")
s_last, s0, L = inp["X_ref"][0, -1], inp["x_ic"][0], tr["L"]
kk = (s0 - s_last).abs() + L / 2
q = torch.stack([s_last + (kk - torch.fmod(kk, L)) * torch.sign(s0 - s_last), inp["X_ref"][1, -1]]).contiguous()
sx5, sj5, _ = sv5.ss_query(q)
o64 = run(sv5, inp, sx5, sj5, 160, B, False)
o32 = run(sv5, inp, sx5, sj5, 160, B, True)
both = (o64["status"] == 0) & (o32["status"] == 0)
e, ed = per_problem_err({k: o32[k][..., both] for k in ("X_optm", "U_optm", "dU_optm")}, {k: o64[k][..., both] for k in ("X_optm", "U_optm", "dU_optm")})
print("S=160 batch 4096: fp64 solved", (o64["status"] == 0).mean(), "mixed solved", (o32["status"] == 0).mean(), "iters", o64["iters"].mean(), o32["iters"].mean(),
      "mixed vs fp64: median", np.median(e), "p99", np.percentile(e, 99), "max", e.max())
