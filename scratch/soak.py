import sys, numpy as np, torch, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
pkg = load_package()
tab = pkg.workloads.track_from_file(ROOT / "tests/golden/barc_track/15_barc_optm.txt", 1024)
N, B = 20, 4096
solver = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), 0)
rng = np.random.default_rng(3)
s0 = rng.uniform(0, tab["L"], B)
x0 = np.stack([s0, rng.uniform(-0.08, 0.08, B), rng.normal(0, 0.03, B), rng.uniform(0.6, 0.95, B) * np.interp(s0, np.arange(1024) * tab["L"] / 1024, tab["vel"]), np.zeros(B), np.zeros(B)])
t0 = time.time()
steps = int(3.2 * tab["L"] / 3.0 / 0.025)
res = pkg.closed_loop.run(solver, tab, torch.as_tensor(x0, device="cuda"), torch.zeros((2, B), dtype=torch.float64, device="cuda"), steps=steps, speed_scale=0.9, graph=True)
torch.cuda.synchronize()
dt = time.time() - t0
d = res["distance"].cpu().numpy(); e = res["worst_excess"].cpu().numpy(); f = res["n_fail"].cpu().numpy()
print("%d cars x %d steps in %.1f s (%.2f M car-steps/s); laps min %.2f median %.2f; cars outside by >1 cm: %d (max %.3f m); cars with a failed solve: %d (max %d fails); non-finite states: %d" % (B, steps, dt, B * steps / dt / 1e6, d.min() / tab["L"], np.median(d) / tab["L"], (e > 0.01).sum(), np.nanmax(e), (f > 0).sum(), f.max(), (~np.isfinite(res["x"].cpu().numpy())).sum()))
