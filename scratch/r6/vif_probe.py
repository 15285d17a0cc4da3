"""Round 6: calibrate the conditioning test of the fp32 polish (probe build: kkt[0] = variance inflation, kkt[1] = max diag C_A^-1)."""
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
K = ("X_optm", "U_optm", "dU_optm")


def batch(kind, B, seed):
    laps = DC.spec_laps() if kind == "spec" else pkg.workloads.synthetic_laps(tr, 5)
    cfgd = dict(pkg.presets.barc_lmpc(20, 5)); cfgd["polish"] = 1      # marks visible, no second pass
    sv = pkg.Solver(cfgd, pkg.presets.barc_vehicle(), device=0)
    sv.set_safe_set(laps, tr["L"])
    if kind == "spec":
        x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], seed=seed)
    else:
        x, u = pkg.workloads.sample_states_near_laps(laps, B, tr["L"], seed=seed)
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
    ver = (om["status"] == 0) & (o64["status"] == 0)
    e, ed = per_problem_err({k: om[k] for k in K}, {k: o64[k] for k in K})
    vif, cinv = om["kkt"][0], om["kkt"][1]
    print("== %s seed %d: %d problems, verified by the fp32 pass %d, marks %d" % (kind, seed, B, ver.sum(), (om["status"] == 3).sum()))
    for name, v in (("vif", vif), ("cinv", cinv)):
        qs = np.quantile(v[ver], [0.5, 0.9, 0.99, 0.999, 1.0])
        print("   %s of verified: median %.2e 90%% %.2e 99%% %.2e 99.9%% %.2e max %.2e" % (name, *qs))
    for thr in (1e-4, 3e-4, 1e-3):
        m = ver & (e > thr)
        print("   e > %.0e: %d problems; their vif min %.2e, cinv min %.2e" % (thr, m.sum(), vif[m].min() if m.any() else 0, cinv[m].min() if m.any() else 0))
    for vt in (10, 30, 100, 300, 1000):
        m = ver & (vif > vt)
        print("   vif > %5d: %5d verified problems (%.2f %%), worst error left below it %.1e" % (vt, m.sum(), 100 * m.sum() / ver.sum(), e[ver & ~m].max()))
    for ct in (1, 3, 10, 30, 100, 1000):
        m = ver & (cinv > ct)
        print("   cinv > %5d: %5d verified problems (%.2f %%), worst error left below it %.1e" % (ct, m.sum(), 100 * m.sum() / ver.sum(), e[ver & ~m].max()))
    w = np.argsort(np.where(ver, e, 0))[-10:][::-1]
    print("   worst:", [(int(b), "%.1e" % e[b], "vif %.1e" % vif[b], "cinv %.1e" % cinv[b]) for b in w])
    sv.close()


batch("spec", 32768, 0)
batch("spec", 32768, 1)
batch("near", 32768, 0)
