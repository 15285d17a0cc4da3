import sys, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
pkg = load_package()
tr = pkg.racing_trajectory.RacingTrajectory(ROOT / "tests/golden/barc_track/15_barc_optm.txt")
tab = tr.to_track_table(1024)
print("L %.3f  vel %.2f..%.2f  |k| max %.2f  half width %.2f..%.2f" % (tab["L"], tab["vel"].min(), tab["vel"].max(), np.abs(tab["curvature"]).max(), tab["bound_left"].min(), tab["bound_left"].max()))
N, B = 20, 256
solver = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), 0)
rng = np.random.default_rng(1)
s0 = rng.uniform(0, tab["L"], B)
x0 = np.stack([s0, rng.uniform(-0.05, 0.05, B), np.zeros(B), 0.8 * np.interp(s0, np.arange(1024) * tab["L"] / 1024, tab["vel"]), np.zeros(B), np.zeros(B)])
for scale in (0.9, 0.8, 0.7):
    res = pkg.closed_loop.run(solver, tab, torch.as_tensor(x0, device="cuda"), torch.zeros((2, B), dtype=torch.float64, device="cuda"), steps=int(2.2 * tab["L"] / 3.0 / 0.025), speed_scale=scale)
    d = res["distance"].cpu().numpy(); e = res["worst_excess"].cpu().numpy(); f = res["n_fail"].cpu().numpy()
    print("scale", scale, "laps min/median %.2f %.2f" % (d.min() / tab["L"], np.median(d) / tab["L"]), "excess max %.3f  cars outside %d" % (np.nanmax(e), (e > 0).sum()), "fails: cars with any %d, max %d" % ((f > 0).sum(), f.max()))
