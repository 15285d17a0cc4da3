"""Twin (and optionally the kernel's output file) against the dense optimum on the UNCLIPPED cold-start sample, with the
dense multipliers' strict-complementarity measure per problem: usage r2_acc.py N [B] [seed]"""
import sys, numpy as np, time
from multiprocessing import Pool
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "racing-lmpc-ros2_amd"))
import workloads as wl
from oracle import cbind, params as P, qp as Q, scenario as S

def work(args):
    N, b, kind = args
    cfg, veh, inp = G[0], G[1], G[2]
    qp = Q.build_qp(cfg, veh, S.problem(inp, b))
    y, info = Q.solve_dense(qp)
    ex = qp.split(y)
    slack = qp.d - qp.C @ y
    sc = Q.strict_complementarity(qp, y, info["lam"])
    return b, info["status"], bool(info.get("polished")), ex["X_optm"], ex["U_optm"], ex["dU_optm"], sc

if __name__ == "__main__":
    N = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 256; seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    kind = sys.argv[4] if len(sys.argv) > 4 else "barc"
    veh = P.barc_vehicle() if kind == "barc" else P.iac_vehicle()
    cfg = P.barc_tracking_mpc(N) if kind == "barc" else P.iac_tracking_mpc(N)
    tr = wl.synthetic_track(kind)
    u_lo, u_hi = Q.effective_bounds(cfg, veh)[:2]
    x, u = wl.sample_initial_states(kind, B, tr["L"], u_lo, u_hi, seed=seed)
    inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
    G = (cfg, veh, inp)
    t0 = time.time()
    o = cbind.solve_batch(cfg, veh, inp)
    print("twin status", np.bincount(o["status"], minlength=3), "mean iters", o["iters"].mean())
    import os, pickle
    cache = f"/tmp/dense_{kind}_{N}_{seed}_{B}.pkl"
    if os.path.exists(cache):
        res = pickle.load(open(cache, "rb"))
    else:
        with Pool(8) as pool:
            res = pool.map(work, [(N, b, kind) for b in range(B)])
        pickle.dump(res, open(cache, "wb"))
    tol = float(os.environ.get("TWIN_TOL", "0"))
    if os.environ.get("TWIN_LIB"):
        import ctypes
        cbind._LIB = ctypes.CDLL(os.environ["TWIN_LIB"])
        if not tol: tol = 1e-11
    if tol:
        o = cbind.solve_batch(cfg, veh, inp, tol=tol)
        print("twin tol", tol, "status", np.bincount(o["status"], minlength=3), "mean iters", o["iters"].mean(), "hist", np.bincount(o["iters"]))
    ex_ = np.zeros(B); eu_ = np.zeros(B); sc_ = np.zeros(B); dst = np.zeros(B, int)
    for b, st, pol, X, U, dU, sc in res:
        ex_[b] = (np.abs(o["X_optm"][:, :, b] - X) / P.SCALE_X[:, None]).max()
        eu_[b] = (np.abs(o["U_optm"][:, :, b] - U) / P.SCALE_U[:, None]).max()
        sc_[b] = sc; dst[b] = st
    e = np.maximum(ex_, eu_)
    print(f"N={N} dense status {np.bincount(dst, minlength=3)} time {time.time()-t0:.0f}s")
    print("err quantiles 50/90/99/max", np.quantile(e, [.5, .9, .99, 1.0]))
    for thr in (1e-3, 1e-4, 1e-5, 1e-6):
        strict = sc_ >= thr
        print(f"  strict(sc>={thr:g}): {strict.mean():.3f} of problems; worst err strict {e[strict].max() if strict.any() else 0:.2e}; worst err degenerate {e[~strict].max() if (~strict).any() else 0:.2e}")
    worst = np.argsort(-e)[:10]
    for b in worst: print(f"   b={b} err {e[b]:.1e} sc {sc_[b]:.1e} st {o['status'][b]} it {o['iters'][b]} vx0 {x[b,3]:.2f}")
    np.savez(f"/tmp/acc_{kind}_{N}_{seed}.npz", e=e, sc=sc_, st=o["status"], it=o["iters"])
