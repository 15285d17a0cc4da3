import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
pkg = load_package()
import importlib
capi = importlib.import_module(pkg.__name__ + ".capi")
_orig = capi.library_path
capi.library_path = lambda: _orig().with_name(os.environ.get("LMPC_LIB", "liblmpc_hip.so"))
N = int(sys.argv[1]); B = 256
tr = pkg.workloads.synthetic_track("barc")
solver = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), device=0)
x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.314], [0.01, 0.314], seed=0)
inp = solver.prepare(tr, x.T.copy(), 0.025)
inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device="cuda")
res = []
for rep in range(int(os.environ.get("REPS", "8"))):
    o = solver.solve(inp)
    res.append((o["status"].cpu().numpy().copy(), o["X_optm"].cpu().numpy().copy()))
print(os.environ.get("LMPC_LIB"), "tracking N", N, "unsolved", sum(int((r[0] != 0).sum()) for r in res), "bitwise diffs vs rep 0", sum(int((r[1] != res[0][1]).any(axis=(0, 1)).sum()) for r in res[1:]))
