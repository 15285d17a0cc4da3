"""Polish prototype (scratch/r3_polish_oracle.c) against the dense optimum on the unclipped cold-start sample.
usage: r3_pol_eval.py N [B] [seed] [kind]   env: POL_* knobs, TOLS="3e-14,1e-9"."""
import sys, os, ctypes, pickle, numpy as np
from multiprocessing import Pool
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "racing-lmpc-ros2_amd"))
import workloads as wl
from oracle import cbind, params as P, qp as Q, scenario as S

def work(args):
    N, b, kind = args
    cfg, veh, inp = G[:3]
    kw = {} if len(G) == 3 else {"ss_x": G[3][:, :, b], "ss_j": G[4][:, b]}
    qp = Q.build_qp(cfg, veh, S.problem(inp, b), **kw)
    y, info = Q.solve_dense(qp)
    ex = qp.split(y)
    sc = Q.strict_complementarity(qp, y, info["lam"])
    return b, info["status"], bool(info.get("polished")), ex["X_optm"], ex["U_optm"], ex["dU_optm"], sc

if __name__ == "__main__":
    N = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 256; seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    kind = sys.argv[4] if len(sys.argv) > 4 else "barc"
    ss_x = ss_j = None
    if kind == "lmpc":
        sys.path.insert(0, str(ROOT / "tests"))
        import lmpc_scenario as LS
        veh, cfg, tr, laps, inp, q = LS.make(B, seed, N)
        ss_x, ss_j, _ = LS.oracle_safe_set(cfg, laps, q)
        G = (cfg, veh, inp, ss_x, ss_j)
    elif kind == "lmpc160":
        veh, cfg = P.barc_vehicle(), P.barc_lmpc(N, 5)
        tr = wl.synthetic_track("barc"); laps = wl.synthetic_laps(tr, 5)
        x, u = wl.sample_states_near_laps(laps, B, tr["L"], seed=seed)
        inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
        from oracle import dynamics as D
        q = np.stack([D.align_abscissa(inp["X_ref"][0, -1, :], inp["x_ic"][0, :], tr["L"]), inp["X_ref"][1, -1, :]])
        ss_x, ss_j, _ = cbind.ss_query_batch(laps[-cfg.max_lap_stored:], tr["L"], cfg.num_ss_pts, cfg.num_ss_pts_per_lap, q)
        G = (cfg, veh, inp, ss_x, ss_j)
    else:
        veh = P.barc_vehicle() if kind == "barc" else P.iac_vehicle()
        cfg = P.barc_tracking_mpc(N) if kind == "barc" else P.iac_tracking_mpc(N)
        tr = wl.synthetic_track("barc" if kind == "barc" else "putnam")
        u_lo, u_hi = Q.effective_bounds(cfg, veh)[:2]
        x, u = wl.sample_initial_states("barc" if kind == "barc" else "putnam", B, tr["L"], u_lo, u_hi, seed=seed)
        inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
        G = (cfg, veh, inp)
    cache = f"/tmp/dense_{kind}_{N}_{seed}_{B}.pkl"
    if os.path.exists(cache):
        res = pickle.load(open(cache, "rb"))
    else:
        with Pool(8) as pool:
            res = pool.map(work, [(N, b, kind) for b in range(B)])
        pickle.dump(res, open(cache, "wb"))
    lib = ctypes.CDLL(os.environ.get("TWIN_LIB", "/tmp/liboracle_pol.so"))
    cbind._LIB = lib
    stats = (ctypes.c_int * 8).in_dll(lib, "g_pol_stats")
    sc_ = np.array([r[6] for r in res]); pol_ = np.array([r[2] for r in res]); dst = np.array([r[1] for r in res])
    print(f"{kind} N={N} B={B}: dense status {np.bincount(dst, minlength=3)}, dense polished {pol_.mean():.3f}, degenerate (sc<1e-4) {np.mean(sc_ < 1e-4):.3f}")
    for tol in [float(t) for t in os.environ.get("TOLS", "3e-14").split(",")]:
        for k in range(8): stats[k] = 0
        o = cbind.solve_batch(cfg, veh, inp, ss_x=ss_x, ss_j=ss_j, tol=tol)
        e = np.zeros(B); ed = np.zeros(B)
        for b, st, pol, X, U, dU, sc in res:
            e[b] = max((np.abs(o["X_optm"][:, :, b] - X) / P.SCALE_X[:, None]).max(), (np.abs(o["U_optm"][:, :, b] - U) / P.SCALE_U[:, None]).max())
            ed[b] = (np.abs(o["dU_optm"][:, :, b] - dU) / P.SCALE_U[:, None]).max()
        okd = (dst == 0)
        print(f" tol {tol:g}: status {np.bincount(o['status'], minlength=3)} iters {o['iters'].mean():.2f}  polish acc/rej/-/skip {list(stats)[:4]} by round {list(stats)[4:]}")
        print(f"   err XU 50/90/99/max {np.quantile(e[okd], [.5, .9, .99, 1.0])}   dU max {ed[okd].max():.2e}  certified-only max {e[okd & pol_].max():.2e}")
        worst = np.argsort(-np.where(okd, e, 0))[:4]
        print("   worst:", [(int(b), f"{e[b]:.1e}", f"sc {sc_[b]:.0e}", int(o['iters'][b])) for b in worst])
