import sys, os, numpy as np, ctypes as C
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package
from oracle import cbind, params as P, qp as Q, scenario as S
pkg = load_package()
_real = C.CDLL(str(ROOT / "scratch/_exp_oracle2.so")); cbind.lib = lambda: _real
d = np.load("gpurun_out/iac_status.npz")
veh, cfg = P.iac_vehicle(), P.iac_tracking_mpc(40)
tr = pkg.workloads.synthetic_track("putnam")
idx = np.where(d["st"] == 2)[0]
x, u = d["x"][idx], d["u"][idx]
inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
for mi in (30, 60):
    tw = cbind.solve_batch(cfg, veh, inp, max_iter=mi)
    print("ratio", os.environ.get("X_RATIO"), "max_iter", mi, "status", np.bincount(tw["status"], minlength=3), "iters hist", np.bincount(tw["iters"]))
