import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from __graft_entry__ import load_package
pkg = load_package()
import importlib, os
capi = importlib.import_module(pkg.__name__ + ".capi")
_orig = capi.library_path
capi.library_path = lambda: _orig().with_name(os.environ.get("LMPC_LIB", "liblmpc_hip.so"))
from oracle import cbind, params as P
import lmpc_scenario as LS
N, n_laps = int(sys.argv[1]), int(sys.argv[2])
veh, cfg, tr, laps, inp, q = LS.make(32, 70 + N, N=N, n_laps=min(n_laps, 3))
cfg = P.barc_lmpc(N, n_laps)
stored = (laps * 2)[:n_laps]
solver = pkg.Solver(pkg.presets.barc_lmpc(N, n_laps), pkg.presets.barc_vehicle(), device=0)
solver.set_safe_set(stored, LS.L_BARC_SS)
ss_x, ss_j, nf = solver.ss_query(q)
res = []
for rep in range(int(os.environ.get('REPS', '3'))):
    out = solver.alloc_outputs(32)
    out["convex_combi_optm"] = torch.zeros((32 * n_laps, 32), dtype=torch.float64, device="cuda")
    o = {k: v.cpu().numpy() for k, v in solver.solve(inp, out, ss_x=ss_x, ss_j=ss_j).items() if hasattr(v, "cpu")}
    res.append((o["status"].copy(), o["iters"].copy(), o["X_optm"].copy()))

nbad = sum(int((r[0] != 0).sum()) for r in res)
ndiff = sum(int((r[2] != res[0][2]).any(axis=(0, 1)).sum()) for r in res[1:])
print(os.environ.get("LMPC_LIB"), "reps", len(res), "unsolved total", nbad, "problems differing bitwise from rep 0 (summed)", ndiff)
