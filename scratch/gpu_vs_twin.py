import sys, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package
from oracle import cbind, params as P, qp as Q, scenario as S
pkg = load_package()
veh, cfg = P.barc_vehicle(), P.barc_tracking_mpc(20)
tr = pkg.workloads.synthetic_track("barc")
u_lo, u_hi, _, _ = Q.effective_bounds(cfg, veh)
B = 48
x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], u_lo, u_hi, 21)
inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
tw = cbind.solve_batch(cfg, veh, inp)
solver = pkg.Solver(pkg.presets.barc_tracking_mpc(20), pkg.presets.barc_vehicle(), 0)
out = {k: v.cpu().numpy() for k, v in solver.solve(inp).items() if hasattr(v, "cpu")}
ex = np.abs((out["X_optm"] - tw["X_optm"]) / P.SCALE_X[:, None, None]).max(axis=(0, 1))
print("iters gpu", out["iters"]); print("iters twin", tw["iters"])
print("err vs twin per problem", np.array2string(ex, precision=1))
print("mu gpu", np.array2string(out["kkt"][2], precision=1)); print("mu twin", np.array2string(tw["kkt"][2], precision=1))
