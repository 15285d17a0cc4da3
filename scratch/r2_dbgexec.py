import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from __graft_entry__ import load_package
pkg = load_package()
import importlib
capi = importlib.import_module(pkg.__name__ + ".capi")
_orig = capi.library_path
capi.library_path = lambda: _orig().with_name("liblmpc_hip_dbgexec.so")
from oracle import params as P
import lmpc_scenario as LS
N, n_laps = 80, 5
veh, cfg, tr, laps, inp, q = LS.make(32, 70 + N, N=N, n_laps=3)
solver = pkg.Solver(pkg.presets.barc_lmpc(N, n_laps), pkg.presets.barc_vehicle(), device=0)
solver.set_safe_set((laps * 2)[:n_laps], LS.L_BARC_SS)
ss_x, ss_j, nf = solver.ss_query(q)
for rep in range(3):
    out = solver.alloc_outputs(32)
    out["kkt"] = torch.zeros((20, 32), dtype=torch.float64, device="cuda")
    out["convex_combi_optm"] = torch.zeros((32 * n_laps, 32), dtype=torch.float64, device="cuda")
    o = solver.solve(inp, out, ss_x=ss_x, ss_j=ss_j)
    k = o["kkt"].cpu().numpy()
    print("status", o["status"].cpu().numpy()[:16], "bwd mismatches", k[4, :16], "fwd mismatches", k[5, :16])
