import sys, os, numpy as np, torch, ctypes
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from __graft_entry__ import load_package
pkg = load_package()
from oracle import cbind, params as P
import lmpc_scenario as LS
# swap in the debug build
import importlib
capi = sys.modules[pkg.__name__ + ".capi"] if (pkg.__name__ + ".capi") in sys.modules else importlib.import_module(pkg.__name__ + ".capi")
capi._LIB = None
orig = capi.library_path
capi.library_path = lambda: orig().with_name(os.environ.get("LMPC_LIB", "liblmpc_hip_dbg.so"))
veh, cfg, tr, laps, inp, q = LS.make(2048, 9)
solver = pkg.Solver(pkg.presets.barc_lmpc(20, 3), pkg.presets.barc_vehicle(), device=0)
solver.set_safe_set(laps, LS.L_BARC_SS)
ss_x, ss_j, nf = solver.ss_query(q)
out = solver.alloc_outputs(2048)
out["convex_combi_optm"] = torch.zeros((96, 2048), dtype=torch.float64, device="cuda")
o = {k: v.cpu().numpy() for k, v in solver.solve(inp, out, ss_x=ss_x, ss_j=ss_j).items() if hasattr(v, "cpu")}
lam = o["convex_combi_optm"]
bad = np.where(np.abs(lam.sum(0) - 1) > 1e-8)[0]
print("status", np.bincount(o["status"], minlength=3), "bad lambda-sum:", len(bad), bad[:10], "m max hist", np.bincount(o["kkt"][0].astype(int)))
print("m of bad", o["kkt"][0][bad][:10], "sum err", (lam.sum(0) - 1)[bad][:10], "iters", o["iters"][bad][:10])
sub = {k: (v[..., bad[:8]] if isinstance(v, np.ndarray) else v) for k, v in inp.items()}
tw = cbind.solve_batch(cfg, veh, sub, ss_x=ss_x.cpu().numpy()[..., bad[:8]], ss_j=ss_j.cpu().numpy()[..., bad[:8]])
print("twin on bad: status", tw["status"], "iters", tw["iters"], "X diff", np.abs((o["X_optm"][:, :, bad[:8]] - tw["X_optm"]) / P.SCALE_X[:, None, None]).max(axis=(0, 1)))
for b in bad[:3]:
    print(" support kernel", np.where(lam[:, b] > 1e-6)[0], lam[lam[:, b] > 1e-6, b].round(5), "twin", np.where(tw["convex_combi_optm"][:, list(bad[:8]).index(b)] > 1e-6)[0])
