import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from __graft_entry__ import load_package
pkg = load_package()
import importlib
capi = importlib.import_module(pkg.__name__ + ".capi")
_orig = capi.library_path
capi.library_path = lambda: _orig().with_name(os.environ.get("LMPC_LIB", "liblmpc_hip.so"))
import lmpc_scenario as LS
veh, cfg, tr, laps, inp, q = LS.make(2048, 9)
solver = pkg.Solver(pkg.presets.barc_lmpc(20, 3), pkg.presets.barc_vehicle(), device=0)
solver.set_safe_set(laps, LS.L_BARC_SS)
ss_x, ss_j, nf = solver.ss_query(q)
res = []
for k in range(6):
    out = solver.alloc_outputs(2048)
    out["convex_combi_optm"] = torch.zeros((96, 2048), dtype=torch.float64, device="cuda")
    o = solver.solve(inp, out, ss_x=ss_x, ss_j=ss_j)
    torch.cuda.synchronize()
    res.append((o["X_optm"].clone(), o["iters"].clone(), o["convex_combi_optm"].clone()))
for k in range(1, 6):
    dx = (res[k][0] != res[0][0]).any(dim=0).any(dim=0)
    print("run", k, "problems differing bitwise in X:", int(dx.sum()), "iters differ:", int((res[k][1] != res[0][1]).sum()), "max |dX|", float((res[k][0] - res[0][0]).abs().max()), "lam sum err max", float((res[k][2].sum(0) - 1).abs().max()))
lam = res[0][2].cpu().numpy(); X = res[0][0].cpu().numpy()
err = np.abs(lam.sum(0) - 1)
bad = np.argsort(-err)[:5]
print("worst lambda-sum problems", bad, err[bad], "status", o["status"].cpu().numpy()[bad], "iters", res[0][1].cpu().numpy()[bad], "kkt", o["kkt"].cpu().numpy()[:, bad].T)
from oracle import cbind, params as P
sub = {k: (v[..., bad] if isinstance(v, np.ndarray) else v) for k, v in inp.items()}
tw = cbind.solve_batch(cfg, veh, sub, ss_x=ss_x.cpu().numpy()[..., bad], ss_j=ss_j.cpu().numpy()[..., bad])
print("twin status", tw["status"], "iters", tw["iters"], "X diff", np.abs((X[:, :, bad] - tw["X_optm"]) / P.SCALE_X[:, None, None]).max(axis=(0, 1)))
print("lambda kernel (nonzero)", [(j, round(lam[j, bad[0]], 4)) for j in np.where(np.abs(lam[:, bad[0]]) > 1e-6)[0]], "twin", [(j, round(tw["convex_combi_optm"][j, 0], 4)) for j in np.where(tw["convex_combi_optm"][:, 0] > 1e-6)[0]])
