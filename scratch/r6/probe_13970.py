"""Round 6: problem 13970 of the spec learning batch with the regression on -- MAX_ITER in the kernel, solved by the dense oracle."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package
import dense_cases as DC
from oracle import params as P, qp as Q, scenario as S
pkg = load_package()
import test_gpu_spec_workload as T
B = 32768
sv, tr, inp, ss_x, ss_j = T._setup(pkg, B, True)
o = T._solve(sv, inp, ss_x, ss_j, False)
bad = np.where(o["status"] != 0)[0]
print("bad", bad, o["status"][bad], o["iters"][bad])
b = 13970
A, Bm, g = sv.linearize(inp); sv.regress(inp, A, Bm, g)
A, Bm, g = A.cpu().numpy(), Bm.cpu().numpy(), g.cpu().numpy()
npinp = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in inp.items()}
sx, sj = ss_x.cpu().numpy(), ss_j.cpu().numpy()
cfg, veh = P.barc_lmpc(20, 5), P.barc_vehicle()
lin = (np.ascontiguousarray(A[..., b].transpose(2, 0, 1)), np.ascontiguousarray(Bm[..., b].transpose(2, 0, 1)), np.ascontiguousarray(g[..., b].T))
qp = Q.build_qp(cfg, veh, S.problem(npinp, b), ss_x=sx[:, :, b], ss_j=sj[:, b], lin=lin)
y, info = Q.solve_dense(qp)
ex = qp.split(y)
print("dense status", info["status"], "iters", info["iters"], "polished", info.get("polished"), "objective", qp.objective(y))
c = Q.kkt_certificate(qp, y); print("certificate", c)
lam = ex["convex_combi_optm"]; print("dense support", np.where(lam > 1e-9)[0], lam[lam > 1e-9].round(5), "sigma", ex.get("sigma"))
print("kernel X err vs dense", np.abs((o["X_optm"][:, :, b] - ex["X_optm"]) / P.SCALE_X[:, None]).max(), "kkt", o["kkt"][:, b] if "kkt" in o else None)
lk = o["convex_combi_optm"][:, b]; print("kernel support", np.where(lk > 1e-9)[0], lk[lk > 1e-9].round(5))
print("spectral radius of A per stage:", [round(float(np.abs(np.linalg.eigvals(lin[0][i])).max()), 2) for i in range(19)])
# the same problem alone, more iterations / no polish
for name, kw in (("max_iter 100", dict(max_iter=100)), ("polish off", dict(polish=-1)), ("polish off, 100", dict(polish=-1, max_iter=100))):
    cf = dict(pkg.presets.barc_lmpc(20, 5)); cf.update(kw)
    s1 = pkg.Solver(cf, pkg.presets.barc_vehicle(), device=0)
    s1.set_safe_set(DC.spec_laps(), tr["L"])
    # regression as in the batch
    import types
    pv = dict(pkg.presets.barc_vehicle()); pv["mu"] *= 0.85
    plant = pkg.Solver(cf, pv, device=0)
    reg_laps = pkg.workloads.regression_sample_pairs(tr, DC.spec_laps(), lambda xa, ua: plant.plant_step(tr, torch.as_tensor(xa.T.copy(), device="cuda"), torch.as_tensor(ua.T.copy(), device="cuda"), 0.03).cpu().numpy().T)
    plant.close()
    s1.set_regression_laps(reg_laps, dist_max=0.6)
    i1 = {k: (v[..., b:b + 1].contiguous() if hasattr(v, "dim") and v.dim() >= 1 else v) for k, v in inp.items()}
    out = s1.alloc_outputs(1); out["convex_combi_optm"] = torch.zeros((160, 1), dtype=torch.float64, device="cuda")
    r = s1.solve(i1, out, ss_x=ss_x[:, :, b:b + 1].contiguous(), ss_j=ss_j[:, b:b + 1].contiguous())
    print(name, "status", int(r["status"][0]), "iters", int(r["iters"][0]), "kkt", r["kkt"][:, 0].cpu().numpy(), "X err vs dense",
          np.abs((r["X_optm"][:, :, 0].cpu().numpy() - ex["X_optm"]) / P.SCALE_X[:, None]).max())
    s1.close()
