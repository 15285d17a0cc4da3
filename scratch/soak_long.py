"""Long closed-loop run as a HIP graph: 4096 cars x 10000 control periods (250 s of driving each) on the reference's BARC track."""
import sys, numpy as np, torch, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
pkg = load_package()
tab = pkg.workloads.track_from_file(ROOT / "tests/golden/barc_track/15_barc_optm.txt", 1024)
N, B, steps = 20, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 10000
rng = np.random.default_rng(3)
s0 = rng.uniform(0, tab["L"], B)
x0 = np.stack([s0, rng.uniform(-0.08, 0.08, B), rng.normal(0, 0.03, B), rng.uniform(0.6, 0.95, B) * np.interp(s0, np.arange(1024) * tab["L"] / 1024, tab["vel"]), np.zeros(B), np.zeros(B)])
solver = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), 0)
torch.cuda.synchronize(); t0 = time.time()
r = pkg.closed_loop.run(solver, tab, torch.as_tensor(x0, device="cuda"), torch.zeros((2, B), dtype=torch.float64, device="cuda"), steps=steps, speed_scale=0.9, graph=True)
torch.cuda.synchronize(); dt = time.time() - t0
d = r["distance"].cpu().numpy(); e = r["worst_excess"].cpu().numpy(); f = r["n_fail"].cpu().numpy(); x = r["x"].cpu().numpy()
print("%d cars x %d periods in %.1f s (%.2f M car-steps/s); laps min %.1f median %.1f; non-finite states %d; cars ever outside by >1 cm: %d (max %.3f m); failed solves: %d of %d (%.4f %%), worst car %d" % (
    B, steps, dt, B * steps / dt / 1e6, d.min() / tab["L"], np.median(d) / tab["L"], (~np.isfinite(x)).sum(), (e > 0.01).sum(), np.nanmax(e), f.sum(), B * steps, 100.0 * f.sum() / (B * steps), f.max()))
