"""Round 5: the closed loop (reference's BARC track, HIP graph) with everything behind the solve in one launch
(lmpc_loop_advance_batch) against the ~45 launches it replaces: car-steps/s cold and warm, by batch size."""
import sys, numpy as np, torch, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
pkg = load_package()
tab = pkg.workloads.track_from_file(ROOT / "tests/golden/barc_track/15_barc_optm.txt", 1024)


def states(B):
    rng = np.random.default_rng(3)
    s0 = rng.uniform(0, tab["L"], B)
    return np.stack([s0, rng.uniform(-0.08, 0.08, B), rng.normal(0, 0.03, B), rng.uniform(0.6, 0.95, B) * np.interp(s0, np.arange(1024) * tab["L"] / 1024, tab["vel"]),
                     np.zeros(B), np.zeros(B)])


def loop(N, B, steps, warm, fused, graph=True):
    x0 = states(B)
    solver = pkg.Solver(dict(pkg.presets.barc_tracking_mpc(N)), pkg.presets.barc_vehicle(), 0)
    best = None
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        r = pkg.closed_loop.run(solver, tab, torch.as_tensor(x0, device="cuda"), torch.zeros((2, B), dtype=torch.float64, device="cuda"), steps=steps, speed_scale=0.9,
                                graph=graph, warm=warm, fused=fused)
        torch.cuda.synchronize(); dt = time.time() - t0
        best = dt if best is None else min(best, dt)
    solver.close()
    return best, r


_ = loop(20, 4096, 20, True, True)
for N, B, steps in ((20, 4096, 666), (20, 16384, 200), (20, 1024, 666), (60, 4096, 200)):
    ref = {}
    for warm in (False, True):
        line = "N = %d, %d cars x %d periods, %s:" % (N, B, steps, "warm" if warm else "cold")
        for fused in (False, True):
            dt, r = loop(N, B, steps, warm, fused)
            line += "  %s %.2f M car-steps/s (%.3f ms per period)" % ("one launch behind the solve" if fused else "separate launches", B * steps / dt / 1e6, dt / steps * 1e3)
            ref[fused] = r
        okc = ((ref[True]["n_fail"] == 0) & (ref[False]["n_fail"] == 0)).cpu().numpy()
        sx = np.array([2000.0, 10.0, 0.1, 80.0, 2.0, 2.0])[:, None]
        err = np.abs((ref[True]["x"].cpu().numpy() - ref[False]["x"].cpu().numpy()) / sx)[:, okc].max()
        print(line + "; final states against each other %.1e (scaled); same cars failed: %s; accepted %s"
              % (err, torch.equal(ref[True]["n_fail"] > 0, ref[False]["n_fail"] > 0), ref[True]["warm_hit_rate"]), flush=True)
dt, r = loop(20, 4096, 666, True, True, graph=False)
print("N = 20, 4096 cars, warm, one launch behind the solve, eager (no graph): %.2f M car-steps/s (%.3f ms per period)" % (4096 * 666 / dt / 1e6, dt / 666 * 1e3))
