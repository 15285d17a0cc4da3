"""The twin (polish on, default tolerances) against the dense certified optimum on fresh unclipped samples: every problem.
usage: r3_acc_dense.py [B]   (CPU only; 16 worker processes for the dense solves)"""
import os, sys
os.environ.setdefault("OMP_NUM_THREADS", "1"); os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np, time
from pathlib import Path
from concurrent.futures import ProcessPoolExecutor
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import importlib
from oracle import cbind, params as P, qp as Q, scenario as S
pkg = importlib.import_module("racing-lmpc-ros2_amd")
import os
os.environ.setdefault("OMP_NUM_THREADS", "1"); os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
B0 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
# (name, track, vehicle, config, N, seed, share of B: the dense solve costs O(N^3))
CASES = [("barc tracking N=20", "barc", P.barc_vehicle, P.barc_tracking_mpc, 20, 11, 1.0), ("barc tracking N=40", "barc", P.barc_vehicle, P.barc_tracking_mpc, 40, 12, 0.25),
         ("iac tracking N=40", "putnam", P.iac_vehicle, P.iac_tracking_mpc, 40, 14, 0.25), ("barc tracking N=60", "barc", P.barc_vehicle, P.barc_tracking_mpc, 60, 13, 0.125)]
ONLY = sys.argv[2:]
G = {}
def dense(b):
    cfg, veh, inp = G["cfg"], G["veh"], G["inp"]
    qp = Q.build_qp(cfg, veh, S.problem(inp, b))
    try:
        y, info = Q.solve_dense(qp)
    except np.linalg.LinAlgError:
        return None
    if info["status"] != 0: return None
    o = qp.split(y)
    return o["X_optm"], o["U_optm"], o["dU_optm"]
for name, track, fveh, fcfg, N, seed, share in CASES:
    if ONLY and not any(o in name for o in ONLY): continue
    B = max(16, int(B0 * share))
    veh, cfg = fveh(), fcfg(N)
    tr = pkg.workloads.synthetic_track(track)
    u_lo, u_hi, _, _ = Q.effective_bounds(cfg, veh)
    x, u = pkg.workloads.sample_initial_states(track, B, tr["L"], u_lo, u_hi, seed)
    inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
    G.update(cfg=cfg, veh=veh, inp=inp)
    t0 = time.time()
    with ProcessPoolExecutor(16) as ex:
        res = list(ex.map(dense, range(B), chunksize=4))
    tw = cbind.solve_batch(cfg, veh, inp)
    have = np.array([r is not None for r in res])
    both = have & (tw["status"] == 0)
    per, ped = [], []
    for b in np.nonzero(both)[0]:
        X, U, dU = res[b]
        per.append(max(np.abs((tw["X_optm"][:, :, b] - X) / P.SCALE_X[:, None]).max(), np.abs((tw["U_optm"][:, :, b] - U) / P.SCALE_U[:, None]).max()))
        ped.append(np.abs((tw["dU_optm"][:, :, b] - dU) / P.SCALE_U[:, None]).max())
    per, ped = np.array(per), np.array(ped)
    print(f"{name}: {B} problems, dense solved {have.sum()}, twin solved {(tw['status'] == 0).sum()}, status agree {(have == (tw['status'] == 0)).mean():.4f}; "
          f"X/U max {per.max():.1e} p99 {np.percentile(per, 99):.1e} median {np.median(per):.1e}; dU max {ped.max():.1e}; mean iterations {tw['iters'][both].mean():.2f}  ({time.time() - t0:.0f} s)", flush=True)
    # where the dense solver gave up and the twin did not: the twin's point through the solver-independent KKT certificate
    cert = []
    for b in np.nonzero(~have & (tw["status"] == 0))[0]:
        qp = Q.build_qp(cfg, veh, S.problem(inp, int(b)))
        c = Q.kkt_certificate(qp, Q.pack(qp, tw["X_optm"][:, :, b], tw["U_optm"][:, :, b], tw["dU_optm"][:, :, b], sigma=tw["kkt"][3, b]))
        gs = max(1.0, float(np.abs(qp.H @ Q.pack(qp, tw["X_optm"][:, :, b], tw["U_optm"][:, :, b], tw["dU_optm"][:, :, b], sigma=tw["kkt"][3, b]) + qp.h).max()))
        cert.append((c["stat"] / gs, c["eq"], c["ineq"], c["comp"]))
    if cert:
        c = np.array(cert)
        print(f"   {len(cert)} problems the dense solver gave up on, twin's point certified: stationarity (relative to |gradient|) max {c[:, 0].max():.1e}, equalities {c[:, 1].max():.1e}, inequalities {c[:, 2].max():.1e}, complementarity {c[:, 3].max():.1e}", flush=True)

# ---- the learning problem on the reference's recorded laps (tests/lmpc_scenario.py): 96 safe-set points, N = 20
if not ONLY or any("learning" in o for o in ONLY):
    import lmpc_scenario as LS
    B = max(16, B0 // 8)
    veh, cfg, tr, laps, inp, q = LS.make(B, 21)
    ss_x, ss_j, _ = LS.oracle_safe_set(cfg, laps, q)
    G.update(cfg=cfg, veh=veh, inp=inp, ss_x=ss_x, ss_j=ss_j)
    def dense_l(b):
        qp = Q.build_qp(G["cfg"], G["veh"], S.problem(G["inp"], b), ss_x=G["ss_x"][:, :, b], ss_j=G["ss_j"][:, b])
        try:
            y, info = Q.solve_dense(qp)
        except np.linalg.LinAlgError:
            return None
        if info["status"] != 0: return None
        o = qp.split(y)
        return o["X_optm"], o["U_optm"], o["dU_optm"]
    t0 = time.time()
    with ProcessPoolExecutor(16) as ex:
        res = list(ex.map(dense_l, range(B), chunksize=4))
    tw = cbind.solve_batch(cfg, veh, inp, ss_x, ss_j)
    have = np.array([r is not None for r in res]); both = have & (tw["status"] == 0)
    per = np.array([max(np.abs((tw["X_optm"][:, :, b] - res[b][0]) / P.SCALE_X[:, None]).max(), np.abs((tw["U_optm"][:, :, b] - res[b][1]) / P.SCALE_U[:, None]).max()) for b in np.nonzero(both)[0]])
    ped = np.array([np.abs((tw["dU_optm"][:, :, b] - res[b][2]) / P.SCALE_U[:, None]).max() for b in np.nonzero(both)[0]])
    print(f"barc learning N=20, 96 points (recorded laps): {B} problems, dense solved {have.sum()}, twin solved {(tw['status'] == 0).sum()}, status agree {(have == (tw['status'] == 0)).mean():.4f}; "
          f"X/U max {per.max():.1e} p99 {np.percentile(per, 99):.1e} median {np.median(per):.1e}; dU max {ped.max():.1e}; mean iterations {tw['iters'][both].mean():.2f}  ({time.time() - t0:.0f} s)", flush=True)
