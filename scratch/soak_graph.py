"""Closed loop, eager periods against the captured HIP graph: same final states, different host cost."""
import sys, numpy as np, torch, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
pkg = load_package()
tab = pkg.workloads.track_from_file(ROOT / "tests/golden/barc_track/15_barc_optm.txt", 1024)
N, B = 20, 4096
rng = np.random.default_rng(3)
s0 = rng.uniform(0, tab["L"], B)
x0 = np.stack([s0, rng.uniform(-0.08, 0.08, B), rng.normal(0, 0.03, B), rng.uniform(0.6, 0.95, B) * np.interp(s0, np.arange(1024) * tab["L"] / 1024, tab["vel"]), np.zeros(B), np.zeros(B)])
steps = int(3.2 * tab["L"] / 3.0 / 0.025)
res = {}
import os
for graph, lf in ((False, False), (True, False), (True, True)):
    preset = dict(pkg.presets.barc_tracking_mpc(N)); preset["polish"] = int(os.environ.get("POLISH", "0"))
    solver = pkg.Solver(preset, pkg.presets.barc_vehicle(), 0)
    torch.cuda.synchronize(); t0 = time.time()
    r = pkg.closed_loop.run(solver, tab, torch.as_tensor(x0, device="cuda"), torch.zeros((2, B), dtype=torch.float64, device="cuda"), steps=steps, speed_scale=0.9, graph=graph, longest_first=lf)
    torch.cuda.synchronize(); dt = time.time() - t0
    d = r["distance"].cpu().numpy(); e = r["worst_excess"].cpu().numpy(); f = r["n_fail"].cpu().numpy()
    res[(graph, lf)] = r["x"].cpu().numpy()
    print("graph=%s longest_first=%s: %d cars x %d steps in %.2f s (%.2f M car-steps/s, %.2f ms per period); laps median %.2f; outside by >1 cm: %d; cars with a failed solve: %d (max %d)" % (graph, lf, B, steps, dt, B * steps / dt / 1e6, dt / steps * 1e3, np.median(d) / tab["L"], (e > 0.01).sum(), (f > 0).sum(), f.max()), flush=True)
print("final states identical:", np.array_equal(res[(False, False)], res[(True, False)]), "max diff", np.abs(res[(False, False)] - res[(True, False)]).max())
print("longest-first final states identical to the default order:", np.array_equal(res[(True, True)], res[(True, False)]))
