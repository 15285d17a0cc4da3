import sys, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package
from oracle import params as P, qp as Q, scenario as S
pkg = load_package()
veh, cfg = P.barc_vehicle(), P.barc_tracking_mpc(20)
solver = pkg.Solver(pkg.presets.barc_tracking_mpc(20), pkg.presets.barc_vehicle(), 0)
tr = pkg.workloads.synthetic_track("barc")
u_lo, u_hi, _, _ = Q.effective_bounds(cfg, veh)
x, u = pkg.workloads.sample_initial_states("barc", 96, tr["L"], u_lo, u_hi, 8)
x[:, 3] = np.clip(x[:, 3], 1.6, 3.0)
inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
for ms in (1, 2, 3, 5, 10, 20):
    o = solver.solve_full_dynamics(inp, max_sqp=ms)
    st = o["status"].cpu().numpy(); mv = o["sqp_move"].cpu().numpy(); it = o["sqp_iters"].cpu().numpy()
    print("max_sqp", ms, "status", np.bincount(st, minlength=3), "sqp_iters hist", np.bincount(it), "move pct 10/50/90/max", np.percentile(mv[np.isfinite(mv)], [10, 50, 90, 100]).round(10))
