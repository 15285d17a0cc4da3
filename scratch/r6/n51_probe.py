"""the N = 51 problem of the dispatch sweep that the fused kernel answers with its interior-point iterate: the polish's trace (LMPC_POLISH_TRACE build)"""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package
pkg = load_package()
d = np.load(ROOT / "scratch" / "r6" / "trk_N51.npz")
j, N, B = 7, 51, 64
sv = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), device=0)
keys = ("x_ic", "u_ic", "X_ref", "U_ref", "T_ref", "bound_left", "bound_right", "curvatures", "vel_ref")
inp = {k: torch.as_tensor(np.repeat(d["in_" + k][..., j:j + 1], B, axis=-1).copy(), dtype=torch.float64, device="cuda") for k in keys}
inp["L"] = float(d["L"])
o = sv.solve(inp)
torch.cuda.synchronize()
print("status", o["status"][:4].tolist(), "iters", o["iters"][:4].tolist(), "kkt", o["kkt"][:, 0].tolist())
e = np.abs((o["dU_optm"][..., 0].cpu().numpy() - d["t_dU_optm"][..., j])).max()
print("dU vs the twin's (unscaled)", e)
