"""Round 5: the closed loop of scratch/soak_graph.py (4096 cars x 666 periods on the reference's BARC track) with the warm start:
hit rate, car-steps/s against the cold loop, final states against the cold loop's."""
import sys, numpy as np, torch, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
pkg = load_package()
tab = pkg.workloads.track_from_file(ROOT / "tests/golden/barc_track/15_barc_optm.txt", 1024)
B = 4096
rng = np.random.default_rng(3)
s0 = rng.uniform(0, tab["L"], B)
x0 = np.stack([s0, rng.uniform(-0.08, 0.08, B), rng.normal(0, 0.03, B), rng.uniform(0.6, 0.95, B) * np.interp(s0, np.arange(1024) * tab["L"] / 1024, tab["vel"]), np.zeros(B), np.zeros(B)])
# (one untimed run first: code-object load, graph capture machinery, workspace growth)
_sv = pkg.Solver(dict(pkg.presets.barc_tracking_mpc(20)), pkg.presets.barc_vehicle(), 0)
pkg.closed_loop.run(_sv, tab, torch.as_tensor(x0, device="cuda"), torch.zeros((2, B), dtype=torch.float64, device="cuda"), steps=20, speed_scale=0.9, graph=True, warm=True)
torch.cuda.synchronize(); _sv.close()
for N in (20, 60):
    steps = int(3.2 * tab["L"] / 3.0 / 0.025) if N == 20 else 200
    res = {}
    for graph, warm, lf in ((True, False, False), (True, True, False), (False, True, False), (True, True, True)):
        solver = pkg.Solver(dict(pkg.presets.barc_tracking_mpc(N)), pkg.presets.barc_vehicle(), 0)
        torch.cuda.synchronize(); t0 = time.time()
        r = pkg.closed_loop.run(solver, tab, torch.as_tensor(x0, device="cuda"), torch.zeros((2, B), dtype=torch.float64, device="cuda"), steps=steps, speed_scale=0.9, graph=graph, warm=warm, longest_first=lf)
        torch.cuda.synchronize(); dt = time.time() - t0
        d = r["distance"].cpu().numpy(); f = r["n_fail"].cpu().numpy()
        res[(graph, warm)] = res.get((graph, warm)) if lf else (r["x"].cpu().numpy(), f)
        print(("N = %d graph=%s warm=%s" + (" longest-first" if lf else "") + ": %d cars x %d periods in %.2f s (%.2f M car-steps/s, %.3f ms per period); warm hit rate %s; laps median %.2f; cars with a failed solve: %d"
               ) % (N, graph, warm, B, steps, dt, B * steps / dt / 1e6, dt / steps * 1e3, ("%.4f" % r["warm_hit_rate"]) if warm else "-", np.median(d) / tab["L"], (f > 0).sum()), flush=True)
        solver.close()
    (xc, fc), (xw, fw) = res[(True, False)], res[(True, True)]
    same = (fc == 0) & (fw == 0)
    sx = np.array([2000.0, 10.0, 0.1, 80.0, 2.0, 2.0])[:, None]
    print("N = %d: final states warm vs cold (cars without a failed solve in either: %d): max scaled difference %.2e; same cars failed: %s; eager warm == graph warm: %s"
          % (N, same.sum(), np.abs((xw - xc) / sx)[:, same].max(), np.array_equal(fc > 0, fw > 0), np.array_equal(res[(False, True)][0], xw)), flush=True)
