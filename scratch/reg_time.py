import sys, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
pkg = load_package()
N, B = 20, 4096
solver = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), 0)
tr = pkg.workloads.synthetic_track("barc")
x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.3], [0.01, 0.3], 0)
inp = solver.prepare(tr, x.T.copy(), 0.025)
laps = []
for l, lx in enumerate(pkg.workloads.synthetic_laps(tr, 5)):
    n = lx.shape[0]
    laps.append((lx, np.zeros((n, 2)), lx[:, 5] / np.maximum(lx[:, 3], 0.1), np.arange(n) * 0.03))
solver.set_regression_laps(laps, dist_max=1.0)
A, Bm, g = solver.linearize(inp)
for _ in range(3):
    solver.regress(inp, A, Bm, g)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    solver.regress(inp, A, Bm, g)
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / 10
print("regression kernel: %.3f ms per batch of %d x %d queries over %d lap samples -> %.1f M queries/s" % (ms, B, N - 1, sum(l[0].shape[0] for l in laps), B * (N - 1) / ms / 1e3))
