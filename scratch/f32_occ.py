import sys, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
pkg = load_package()
N, B = 20, 8192
solver = pkg.Solver(pkg.presets.iac_tracking_mpc(N), pkg.presets.iac_vehicle(), 0)
tr = pkg.workloads.synthetic_track("putnam")
x, u = pkg.workloads.sample_initial_states("putnam", B, tr["L"], [-10.0, -0.314159], [5.0, 0.314159], 1)
inp = solver.prepare(tr, x.T.copy(), 0.025)
inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device="cuda")
inp32 = {k: (v.to(torch.float32).contiguous() if hasattr(v, "to") else v) for k, v in inp.items()}
for name, fn, i in (("f32", solver.solve_f32, inp32), ("f64", solver.solve, inp)):
    out = fn(i)
    for _ in range(3): fn(i, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn(i, out)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 20
    it = out["iters"].cpu().numpy(); st = out["status"].cpu().numpy()
    print(name, "IAC N=20 B=%d: %.3f ms -> %.2f M/s; iters mean %.2f max %d; status %s; per-iteration %.1f us" % (B, ms, B / ms / 1e3, it.mean(), it.max(), np.bincount(st, minlength=3), ms * 1e3 / (it.mean() + 1.3)))
