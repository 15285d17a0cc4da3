import sys, os, ctypes, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from oracle import cbind, params as P
import lmpc_scenario as LS
g = dict(np.load(ROOT / "tests/golden/qp_barc_lmpc_n20.npz"))
cfg, veh = P.barc_lmpc(20, 3), P.barc_vehicle()
laps = LS.load_laps()
ss_x, ss_j, _ = LS.oracle_safe_set(cfg, laps, g["query"])
cbind._LIB = ctypes.CDLL(os.environ.get("TWIN_LIB", "/tmp/liboracle_f32.so"))
o = cbind.solve_batch(cfg, veh, g, ss_x=ss_x, ss_j=ss_j)
ex = np.abs((o["X_optm"] - g["X_optm"]) / P.SCALE_X[:, None, None]).max(axis=(0, 1)); eu = np.abs((o["U_optm"] - g["U_optm"]) / P.SCALE_U[:, None, None]).max(axis=(0, 1))
print("status", o["status"], "iters", o["iters"]); print("err", np.maximum(ex, eu)); print("accepted", o["kkt"][1] < 0)
