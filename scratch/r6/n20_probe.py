"""the headline batch's first problems: statuses / iterations / kkt (and the polish trace with the LMPC_POLISH_TRACE build)"""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package
pkg = load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 64
tr = pkg.workloads.synthetic_track("barc")
sv = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), device=0)
x, u = pkg.workloads.sample_initial_states("barc", 4096, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], seed=0)
inp = sv.prepare(tr, x[:B].T.copy(), 0.025)
inp["u_ic"] = torch.as_tensor(u[:B].T.copy(), dtype=torch.float64, device="cuda")
o = sv.solve(inp)
torch.cuda.synchronize()
print("status", o["status"][:16].tolist(), "iters", o["iters"][:16].tolist())
print("kkt[:, 0]", o["kkt"][:, 0].tolist())
