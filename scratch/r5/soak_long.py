"""Round 5: ten simulated minutes of the batched closed loop (4096 cars x 24000 periods of 25 ms on the reference's BARC track; a period =
linearisation + QP + lmpc_loop_advance_batch, replayed as a HIP graph): cold against warm -- rate, acceptance, failures, where the cars end."""
import sys, numpy as np, torch, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
pkg = load_package()
tab = pkg.workloads.track_from_file(ROOT / "tests/golden/barc_track/15_barc_optm.txt", 1024)
B, steps = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 24000
rng = np.random.default_rng(3)
s0 = rng.uniform(0, tab["L"], B)
x0 = np.stack([s0, rng.uniform(-0.08, 0.08, B), rng.normal(0, 0.03, B), rng.uniform(0.6, 0.95, B) * np.interp(s0, np.arange(1024) * tab["L"] / 1024, tab["vel"]), np.zeros(B), np.zeros(B)])
sx = np.array([2000.0, 10.0, 0.1, 80.0, 2.0, 2.0])[:, None]
res = {}
for warm in (False, True):
    sv = pkg.Solver(dict(pkg.presets.barc_tracking_mpc(20)), pkg.presets.barc_vehicle(), 0)
    pkg.closed_loop.run(sv, tab, torch.as_tensor(x0, device="cuda"), torch.zeros((2, B), dtype=torch.float64, device="cuda"), steps=50, speed_scale=0.9, graph=True, warm=warm)
    torch.cuda.synchronize(); t0 = time.time()
    r = pkg.closed_loop.run(sv, tab, torch.as_tensor(x0, device="cuda"), torch.zeros((2, B), dtype=torch.float64, device="cuda"), steps=steps, speed_scale=0.9, graph=True, warm=warm)
    torch.cuda.synchronize(); dt = time.time() - t0
    f = r["n_fail"].cpu().numpy(); d = r["distance"].cpu().numpy(); e = r["worst_excess"].cpu().numpy()
    res[warm] = (r["x"].cpu().numpy(), f, d)
    print("%s: %d cars x %d periods (%.0f simulated s each) in %.1f s: %.2f M car-steps/s, %.3f ms per period; laps median %.1f (min %.1f max %.1f); failed solves %d of %d (%.1e) on %d cars; "
          "worst excursion of a car body beyond the track edge %.3f m (cars that ever left it: %d); accepted %s; finite states: %s"
          % ("warm" if warm else "cold", B, steps, steps * 0.025, dt, B * steps / dt / 1e6, dt / steps * 1e3, np.median(d) / tab["L"], d.min() / tab["L"], d.max() / tab["L"], f.sum(), B * steps,
             f.sum() / (B * steps), (f > 0).sum(), e.max(), (e > 0).sum(), ("%.4f" % r["warm_hit_rate"]) if warm else "-", bool(np.isfinite(res[warm][0]).all())), flush=True)
    sv.close()
(xc, fc, dc), (xw, fw, dw) = res[False], res[True]
same = (fc == 0) & (fw == 0)
print("cars without a failed solve in either loop: %d; their final states warm against cold: max scaled difference %.1e; distance travelled: max difference %.1e m"
      % (same.sum(), np.abs((xw - xc) / sx)[:, same].max(), np.abs(dw - dc)[same].max()))
