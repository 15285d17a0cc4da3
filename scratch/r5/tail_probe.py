"""Round 5: where the kernel-vs-twin tail of profiles/r04_fullsize_parity.txt comes from.  The twin (same algorithm) on the same
batches on the CPU; the problems whose polish was refused (mu > 0 at the exit) solved densely and compared.
usage: tail_probe.py trk80 | lrn60 | trk20big"""
import os, sys
os.environ.setdefault("OMP_NUM_THREADS", "1"); os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np, time, importlib
from pathlib import Path
from concurrent.futures import ProcessPoolExecutor
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from oracle import cbind, params as P, qp as Q, scenario as S
pkg = importlib.import_module("racing-lmpc-ros2_amd")
what = sys.argv[1]
G = {}
def dense(b):
    kw = {} if G.get("ss_x") is None else {"ss_x": G["ss_x"][:, :, b], "ss_j": G["ss_j"][:, b]}
    qp = Q.build_qp(G["cfg"], G["veh"], S.problem(G["inp"], b), **kw)
    y, info = Q.solve_dense(qp)
    o = qp.split(y)
    return b, info["status"], info.get("polished"), o["X_optm"], o["U_optm"], o["dU_optm"], Q.strict_complementarity(qp, y, info["lam"])
tr = pkg.workloads.synthetic_track("barc")
if what.startswith("trk"):
    N = int(what[3:5]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    cfg, veh = P.barc_tracking_mpc(N), P.barc_vehicle()
    x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], seed=0)
    inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025); ss_x = ss_j = None
else:
    N = int(what[3:5]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    cfg, veh = P.barc_lmpc(N, 5), P.barc_vehicle()
    laps = pkg.workloads.synthetic_laps(tr, 5)
    x, u = pkg.workloads.sample_states_near_laps(laps, B, tr["L"], seed=0)
    inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
    s_last, s0, L = inp["X_ref"][0, -1], inp["x_ic"][0], tr["L"]
    kk = np.abs(s0 - s_last) + L / 2
    q = np.stack([s_last + (kk - np.fmod(kk, L)) * np.sign(s0 - s_last), inp["X_ref"][1, -1]])
    ss_x, ss_j, nf = cbind.ss_query_batch([l["x"] if isinstance(l, dict) else l for l in laps], L, cfg.num_ss_pts, cfg.num_ss_pts_per_lap, q)
G.update(cfg=cfg, veh=veh, inp=inp, ss_x=ss_x, ss_j=ss_j)
t0 = time.time()
tw = cbind.solve_batch(cfg, veh, inp, ss_x, ss_j)
print("twin", time.time() - t0, "s; status", np.bincount(tw["status"], minlength=3), "mean iters", tw["iters"].mean())
mu = tw["kkt"][2]
thr = float(os.environ.get("MU_THR", "1e-15"))
ref = np.nonzero((mu > thr) & (tw["status"] == 0))[0]
print("polish not accepted (mu > 0):", ref.size, "of", B, " mu max", mu.max())
sel = list(ref[:int(sys.argv[3]) if len(sys.argv) > 3 else 64])
with ProcessPoolExecutor(8) as ex:
    res = list(ex.map(dense, sel, chunksize=1))
for b, st, pol, X, U, dU, marg in res:
    ex_ = np.abs((tw["X_optm"][:, :, b] - X) / P.SCALE_X[:, None]).max(); eu = np.abs((tw["U_optm"][:, :, b] - U) / P.SCALE_U[:, None]).max()
    ed = np.abs((tw["dU_optm"][:, :, b] - dU) / P.SCALE_U[:, None]).max()
    if max(ex_, eu, ed) > float(os.environ.get("SHOW", "1e-8")): print(f"b={b} dense st {st} pol {pol} margin {marg:.1e} | twin iters {tw['iters'][b]} mu {mu[b]:.1e} rg {tw['kkt'][0, b]:.1e} | err X {ex_:.1e} U {eu:.1e} dU {ed:.1e}")
print(time.time() - t0, "s")
