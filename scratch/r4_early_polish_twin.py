"""Round 4: when should the early polish attempt happen?  The serial twin (same algorithm as the kernel) over the bench's cold starts
with POLISH_MU / POLISH_RD varied (variants of oracle/c/lmpc_oracle.c built by the caller into /tmp/w/tw with sed + gcc -- the
same way the fraction to the boundary, the Mehrotra exponent and the start-point constants were swept: DESIGN.md section 4): mean / max iteration count (interior-point
iterations + polish rounds, what the kernel's time follows), statuses, distance from the shipped thresholds' answers.  CPU only."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402
from oracle import cbind, params as OP, qp as OQ, scenario as OS  # noqa: E402

pkg = load_package()


def inputs(kind, N, B):
    if kind == "barc":
        veh, cfg, tk = OP.barc_vehicle(), OP.barc_tracking_mpc(N), "barc"
        seed = 0
    else:
        veh, cfg, tk = OP.iac_vehicle(), OP.iac_tracking_mpc(N), "putnam"
        seed = 1
    tr = pkg.workloads.synthetic_track(tk)
    u_lo, u_hi, _, _ = OQ.effective_bounds(cfg, veh)
    x, u = pkg.workloads.sample_initial_states(tk, B, tr["L"], u_lo, u_hi, seed)
    return cfg, veh, OS.cold_start_inputs(cfg, veh, tr, x, u, 0.025)


for kind, N, B in (("barc", 20, 2048), ("barc", 40, 512), ("iac", 40, 1024)):
    cfg, veh, inp = inputs(kind, N, B)
    ref = None
    for mu in ("1e-8", "1e-7", "1e-6", "1e-5", "1e-4"):
        for rd in ("1e-6", "1e-4"):
            cbind._LIB = C.CDLL("/tmp/w/tw/lib_%s_%s.so" % (mu, rd))
            o = cbind.solve_batch(cfg, veh, inp)
            if ref is None:
                ref = o
            ok = (o["status"] == 0) & (ref["status"] == 0)
            e = np.abs((o["X_optm"] - ref["X_optm"]) / OP.SCALE_X[:, None, None]).max(axis=(0, 1))[ok].max()
            print("%s N=%d B=%d  POLISH_MU %s RD %s: iters mean %.3f max %d  status %s  max |dX| vs shipped %.1e" % (
                kind, N, B, mu, rd, o["iters"].mean(), o["iters"].max(), np.bincount(o["status"], minlength=3).tolist(), e), flush=True)
