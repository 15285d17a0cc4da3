"""Iteration counts of the C twin on the bench workload for different start-point parameters (env X_*)."""
import os, sys, numpy as np, ctypes as C
from pathlib import Path
from concurrent.futures import ThreadPoolExecutor
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
from oracle import cbind, params as OP, qp as OQ, scenario as OS
pkg = load_package()
so = sys.argv[1] if len(sys.argv) > 1 else str(ROOT / "oracle" / "_build" / "liblmpc_oracle.so")
cbind._LIB = None
_real = C.CDLL(so)
cbind.lib = lambda: _real
cbind.lib().lmpc_oracle_solve_range.restype = C.c_int
veh, cfg = OP.barc_vehicle(), OP.barc_tracking_mpc(20)
tr = pkg.workloads.synthetic_track("barc")
u_lo, u_hi, _, _ = OQ.effective_bounds(cfg, veh)
B = 1024
x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], u_lo, u_hi, 0)
inp = OS.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
T = 16
outs = [None] * T
def work(c):
    outs[c] = cbind.solve_batch(cfg, veh, inp, b0=c * B // T, b1=(c + 1) * B // T)
with ThreadPoolExecutor(T) as ex:
    list(ex.map(work, range(T)))
it = np.concatenate([o["iters"][c * B // T:(c + 1) * B // T] for c, o in enumerate(outs)])
st = np.concatenate([o["status"][c * B // T:(c + 1) * B // T] for c, o in enumerate(outs)])
X = np.concatenate([o["X_optm"][..., c * B // T:(c + 1) * B // T] for c, o in enumerate(outs)], axis=-1)
print({k: os.environ[k] for k in os.environ if k.startswith("X_")}, "status", np.bincount(st, minlength=3), "iters mean %.2f p90 %d max %d" % (it.mean(), np.percentile(it, 90), it.max()))
np.save("/tmp/exp_X_%s.npy" % "_".join(f"{k}{os.environ[k]}" for k in sorted(os.environ) if k.startswith("X_")), X)
