import sys, os, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import importlib
pkg = importlib.import_module("racing-lmpc-ros2_amd")
from oracle import cbind, params as P
N = int(sys.argv[1]) if len(sys.argv) > 1 else 80
g = dict(np.load(ROOT / f"tests/golden/qp_barc_tracking_long_n{N}.npz"))
twin = cbind.solve_batch(P.barc_tracking_mpc(N), P.barc_vehicle(), g)
for polish in (0, -1):
    cfg = dict(pkg.presets.barc_tracking_mpc(N)); cfg["polish"] = polish
    sv = pkg.Solver(cfg, pkg.presets.barc_vehicle(), device=0)
    outs = []
    for rep in range(3):
        o = sv.solve(g); torch.cuda.synchronize()
        outs.append({k: v.cpu().numpy().copy() for k, v in o.items() if hasattr(v, "cpu")})
    o = outs[0]
    e = np.abs((o["X_optm"] - g["X_optm"]) / P.SCALE_X[:, None, None]).max(axis=(0, 1))
    print(f"N={N} polish={polish}: status {np.bincount(o['status'], minlength=3)} iters {o['iters']} ")
    print("   twin iters", twin["iters"], "twin status", np.bincount(twin["status"], minlength=3))
    print("   err", np.array2string(e, precision=1), "repro:", all(np.array_equal(outs[0]["X_optm"], x["X_optm"]) for x in outs[1:]))
    bad = np.where(o["status"] != 0)[0]
    for b in bad: print("   bad", b, "status", o["status"][b], "iters", o["iters"][b], "kkt", o["kkt"][:, b])
    sv.close()
