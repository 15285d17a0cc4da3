import sys, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package
from oracle import params as P, qp as Q, scenario as S
pkg = load_package()
for name, preset, vehp, N, scale_x, scale_u in (("qp_barc_tracking_n20", "barc_tracking_mpc", "barc_vehicle", 20, P.SCALE_X, P.SCALE_U), ("qp_iac_tracking_n40", "iac_tracking_mpc", "iac_vehicle", 40, P.SCALE_X, P.SCALE_U)):
    g = dict(np.load(ROOT / "tests/golden" / (name + ".npz")))
    solver = pkg.Solver(getattr(pkg.presets, preset)(N), getattr(pkg.presets, vehp)(), 0)
    o64 = {k: v.cpu().numpy() for k, v in solver.solve(g).items() if hasattr(v, "cpu")}
    o32 = {k: v.cpu().numpy() for k, v in solver.solve_f32(g).items() if hasattr(v, "cpu")}
    ex = np.abs((o32["X_optm"] - g["X_optm"]) / scale_x[:, None, None]).max(axis=(0, 1))
    eu = np.abs((o32["U_optm"] - g["U_optm"]) / scale_u[:, None, None]).max(axis=(0, 1))
    print(name, "f32 status", o32["status"], "iters", o32["iters"], "(f64 iters", o64["iters"], ")")
    print("   err X", np.array2string(ex, precision=1), "\n   err U", np.array2string(eu, precision=1), "\n   mu", np.array2string(o32["kkt"][2], precision=1), "rd", np.array2string(o32["kkt"][1], precision=1))
# throughput
for N, B, preset, vehp, kind in ((20, 4096, "barc_tracking_mpc", "barc_vehicle", "barc"), (40, 8192, "iac_tracking_mpc", "iac_vehicle", "putnam")):
    solver = pkg.Solver(getattr(pkg.presets, preset)(N), getattr(pkg.presets, vehp)(), 0)
    tr = pkg.workloads.synthetic_track(kind)
    if kind == "barc":
        x, u = pkg.workloads.sample_initial_states(kind, B, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], 0)
    else:
        x, u = pkg.workloads.sample_initial_states(kind, B, tr["L"], [-10.0, -0.314159], [5.0, 0.314159], 1)
    inp = solver.prepare(tr, x.T.copy(), 0.025)
    inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device="cuda")
    inp32 = {k: (v.to(torch.float32).contiguous() if hasattr(v, "to") else v) for k, v in inp.items()}
    out = solver.solve_f32(inp32)
    o64 = solver.solve(inp)
    for _ in range(3):
        solver.solve_f32(inp32, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        solver.solve_f32(inp32, out)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 20
    st = out["status"].cpu().numpy(); it = out["iters"].cpu().numpy()
    st64 = o64["status"].cpu().numpy()
    ok = (st == 0) & (st64 == 0)
    ex = ((out["X_optm"].double() - o64["X_optm"]).abs() / torch.tensor(P.SCALE_X, device="cuda")[:, None, None]).amax(dim=(0, 1)).cpu().numpy()[ok]
    print("N %d B %d: %.3f ms -> %.2f M solves/s; status %s (f64 %s) iters mean %.2f; |X32 - X64| scaled: median %.1e p90 %.1e p99 %.1e max %.1e" % (N, B, ms, B / ms / 1e3, np.bincount(st, minlength=3), np.bincount(st64, minlength=3), it.mean(), np.median(ex), np.percentile(ex, 90), np.percentile(ex, 99), ex.max()))
