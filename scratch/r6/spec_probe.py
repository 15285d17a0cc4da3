"""Round 6, first GPU call: what the bench's spec LMPC workload looks like problem by problem (configs[2] at 4096, configs[4]'s
share at 32768 with the regression): statuses, the failing problems' indices, mixed against fp64."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package
from parity import per_problem_err
pkg = load_package()
dev = "cuda"
tr = pkg.workloads.synthetic_track("barc")
with np.load(ROOT / "gpurun_out" / "spec_laps.npz") as z:
    laps = [z["lap%d" % i] for i in range(5)]
cfgd = pkg.presets.barc_lmpc(20, 5)
for B, reg in ((4096, False), (32768, True), (32768, False)):
    sv = pkg.Solver(cfgd, pkg.presets.barc_vehicle(), device=0)
    sv.set_safe_set(laps, tr["L"])
    if reg:
        pv = dict(pkg.presets.barc_vehicle()); pv["mu"] *= 0.85
        plant = pkg.Solver(cfgd, pv, device=0)
        reg_laps = pkg.workloads.regression_sample_pairs(tr, laps, lambda xa, ua: plant.plant_step(tr, torch.as_tensor(xa.T.copy(), device=dev), torch.as_tensor(ua.T.copy(), device=dev), 0.03).cpu().numpy().T)
        plant.close()
        sv.set_regression_laps(reg_laps, dist_max=0.6)
    x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], seed=0)
    inp = sv.prepare(tr, x.T.copy(), 0.025)
    inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=dev)
    s_last, s0, L = inp["X_ref"][0, -1], inp["x_ic"][0], tr["L"]
    kk = (s0 - s_last).abs() + L / 2
    q = torch.stack([s_last + (kk - torch.fmod(kk, L)) * torch.sign(s0 - s_last), inp["X_ref"][1, -1]]).contiguous()
    ss_x, ss_j, _ = sv.ss_query(q)
    def solve(mixed):
        out = sv.alloc_outputs(B)
        out["convex_combi_optm"] = torch.zeros((160, B), dtype=torch.float64, device=dev)
        return {k: v.cpu().numpy() for k, v in sv.solve(inp, out, ss_x=ss_x, ss_j=ss_j, mixed=mixed).items() if hasattr(v, "cpu")}
    o64, om = solve(False), solve(True)
    bad64, badm = np.where(o64["status"] != 0)[0], np.where(om["status"] != 0)[0]
    print("B", B, "regression", reg, "fp64 status", np.bincount(o64["status"], minlength=4), "bad", bad64[:20], o64["status"][bad64[:20]], o64["iters"][bad64[:20]])
    print("   mixed status", np.bincount(om["status"], minlength=4), "bad", badm[:20], om["status"][badm[:20]], om["iters"][badm[:20]])
    both = (o64["status"] == 0) & (om["status"] == 0)
    e, ed = per_problem_err({k: om[k][..., both] for k in ("X_optm", "U_optm", "dU_optm")}, {k: o64[k][..., both] for k in ("X_optm", "U_optm", "dU_optm")})
    print("   mixed vs fp64: X/U median %.1e 99.9%% %.1e max %.1e dU max %.1e; iters fp64 %.2f mixed %.2f" % (np.median(e), np.quantile(e, 0.999), e.max(), ed.max(), o64["iters"].mean(), om["iters"].mean()))
    print("   x0 of the failing problems:", x[bad64[:12]].round(3).tolist())
    sv.close()
