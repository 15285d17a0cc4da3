import sys, os, numpy as np, ctypes as C
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from oracle import cbind, params as P
_real = C.CDLL(str(ROOT / "scratch/_exp_oracle2.so")); cbind.lib = lambda: _real
d = np.load("gpurun_out/cl_fail.npz")
inp = {k: d[k] for k in d.files}
veh, cfg = P.barc_vehicle(), P.barc_tracking_mpc(20)
for mi in (30,):
    tw = cbind.solve_batch(cfg, veh, inp, max_iter=mi)
    print("ratio", os.environ.get("X_RATIO"), "status", tw["status"], "iters", tw["iters"], "rd", tw["kkt"][1], "mu", tw["kkt"][2])
