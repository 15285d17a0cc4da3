"""Golden vectors at the SHIPPED horizons on the UNCLIPPED cold-start distribution
(run from the repo root: `python tests/golden/make_golden_long.py [N ...]`, ~10 min on 8 cores).

barc_tracking_mpc.param.yaml ships n = 60, iac_car_tracking_mpc n = 80, barc_lmpc n = 40.  Below ~1 m/s the
RK4-discretised tyre dynamics have |eig A| up to ~15-25 per 25 ms step; the cold-start sample of
workloads.sample_initial_states (vx ~ U[0.5, 3]) contains such starts, and they are the problems a structured
factorisation has to survive.  For each horizon this writes

  qp_barc_tracking_long_n{N}.npz   the first B_FULL problems of the sample (seed 0): solver inputs in the C-ABI layout,
                                   the dense polished optimum of the QP racing_mpc.cpp:106-201 defines (oracle/qp.py),
                                   its KKT certificate, the dense solver's status, and the strict-complementarity
                                   margin of the dense multipliers (oracle/qp.py strict_complementarity), which labels
                                   a problem degenerate independently of the solver under test;
  long_status_n{N}.npz             dense status and margin of ALL B_ALL problems of the sample (inputs are regenerated
                                   from the seed by the test): every problem the dense solver solves must come back
                                   status 0 from the kernel.
"""
import os
import sys
from multiprocessing import Pool
from pathlib import Path

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np  # noqa: E402

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402
from oracle import params as OP, qp as OQ, scenario as OS  # noqa: E402

B_ALL, B_FULL = 256, {40: 64, 60: 64, 80: 48}
SEED = 0
_G = {}


def sample(N, B):
    pkg = load_package()
    veh, cfg = OP.barc_vehicle(), OP.barc_tracking_mpc(N)
    tr = pkg.workloads.synthetic_track("barc")
    u_lo, u_hi, _, _ = OQ.effective_bounds(cfg, veh)
    x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], u_lo, u_hi, SEED)
    return veh, cfg, OS.cold_start_inputs(cfg, veh, tr, x, u, 0.025)


def work(b):
    veh, cfg, inp = _G["v"]
    qp = OQ.build_qp(cfg, veh, OS.problem(inp, b))
    y, info = OQ.solve_dense(qp)
    o = qp.split(y)
    c = OQ.kkt_certificate(qp, y) if b < _G["full"] else {"stat": 0, "eq": 0, "ineq": 0, "comp": 0}
    ok = info["status"] == 0 and bool(info.get("polished"))
    return (b, info["status"], ok, OQ.strict_complementarity(qp, y, info["lam"]), o["X_optm"], o["U_optm"], o["dU_optm"],
            o["sigma"], qp.objective(y), [c["stat"], c["eq"], c["ineq"], c["comp"]])


if __name__ == "__main__":
    for N in [int(a) for a in sys.argv[1:]] or [40, 60, 80]:
        veh, cfg, inp = sample(N, B_ALL)
        _G["v"], _G["full"] = (veh, cfg, inp), B_FULL[N]
        with Pool(min(8, os.cpu_count() or 1)) as pool:
            res = pool.map(work, range(B_ALL))
        st = np.array([r[1] for r in res], dtype=np.int32)
        ok = np.array([r[2] for r in res])
        mg = np.array([r[3] for r in res])
        np.savez_compressed(Path(__file__).parent / f"long_status_n{N}.npz", dense_status=st, certified=ok, margin=mg,
                            seed=SEED, batch=B_ALL)
        F = B_FULL[N]
        arrs = {k: (np.asarray(v)[..., :F] if np.ndim(v) else np.asarray(v)) for k, v in inp.items()}
        np.savez_compressed(Path(__file__).parent / f"qp_barc_tracking_long_n{N}.npz",
                            X_optm=np.stack([r[4] for r in res[:F]], -1), U_optm=np.stack([r[5] for r in res[:F]], -1),
                            dU_optm=np.stack([r[6] for r in res[:F]], -1), sigma=np.array([r[7] for r in res[:F]]),
                            objective=np.array([r[8] for r in res[:F]]), kkt_cert=np.array([r[9] for r in res[:F]]).T,
                            certified=ok[:F], dense_status=st[:F], margin=mg[:F], **arrs)
        print(f"N={N}: dense status {np.bincount(st, minlength=3)}, certified {ok.mean():.3f}, "
              f"degenerate (margin < {OQ.DEGENERATE_MARGIN:g}) {np.mean(mg < OQ.DEGENERATE_MARGIN):.3f}", flush=True)
