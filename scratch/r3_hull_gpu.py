import sys, numpy as np, dataclasses, importlib, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import lmpc_scenario as LS
from oracle import cbind, qp as Q, params as P, scenario as S
pkg = importlib.import_module("racing-lmpc-ros2_amd")
B = 64
veh, cfg, tr, laps, inp, q = LS.make(B, 5)
for W in (0.0, 1e9, 1e10, 1e11, 1e12):
    preset = pkg.presets.barc_lmpc(20, 3)
    preset["convex_hull_slack"] = [W] * 6
    for pol in (0, -1):
        preset["polish"] = pol
        solver = pkg.Solver(preset, pkg.presets.barc_vehicle(), device=0)
        solver.set_safe_set(laps, LS.L_BARC_SS)
        ss_x, ss_j, _ = solver.ss_query(q)
        out = solver.alloc_outputs(B)
        out["convex_combi_optm"] = torch.zeros((96, B), dtype=torch.float64, device="cuda")
        o = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in solver.solve(inp, out, ss_x=ss_x, ss_j=ss_j).items()}
        sx, sj = ss_x.cpu().numpy(), ss_j.cpu().numpy()
        tw = cbind.solve_batch(dataclasses.replace(cfg, convex_hull_slack=np.full(6, W)), veh, inp, sx, sj, polish=pol)
        diff = np.nonzero(o["status"] != tw["status"])[0]
        both = (o["status"] == 0) & (tw["status"] == 0)
        e = np.abs((o["X_optm"][:, :, both] - tw["X_optm"][:, :, both]) / P.SCALE_X[:, None, None]).max(axis=(0, 1))
        print(f"W={W:g} polish={pol}: kernel status {np.bincount(o['status'], minlength=3)} twin {np.bincount(tw['status'], minlength=3)} differ {[(int(b), int(o['status'][b]), int(tw['status'][b]), int(o['iters'][b]), int(tw['iters'][b])) for b in diff]} max diff where both ok {e.max():.1e} iters kernel {o['iters'][both].mean():.1f} twin {tw['iters'][both].mean():.1f}")
        solver.close()
