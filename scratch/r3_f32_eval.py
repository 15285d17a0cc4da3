"""fp32 emulation of the twin (scratch/r3_polish_oracle_f32.c) against the polished fp64 twin on a large batch.
usage: r3_f32_eval.py kind N B seed ; env for the f32 run: POL_* ; the reference always runs with the fp64 polish."""
import sys, os, ctypes, numpy as np, subprocess, pickle
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "racing-lmpc-ros2_amd")); sys.path.insert(0, str(ROOT / "tests"))
import workloads as wl
from oracle import cbind, params as P, qp as Q, scenario as S, dynamics as D

def build(kind, N, B, seed):
    ss_x = ss_j = None
    if kind == "lmpc":
        import lmpc_scenario as LS
        veh, cfg, tr, laps, inp, q = LS.make(B, seed, N)
        ss_x, ss_j, _ = LS.oracle_safe_set(cfg, laps, q)
    elif kind == "lmpc160":
        veh, cfg = P.barc_vehicle(), P.barc_lmpc(N, 5)
        tr = wl.synthetic_track("barc"); laps = wl.synthetic_laps(tr, 5)
        x, u = wl.sample_states_near_laps(laps, B, tr["L"], seed=seed)
        inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
        q = np.stack([D.align_abscissa(inp["X_ref"][0, -1, :], inp["x_ic"][0, :], tr["L"]), inp["X_ref"][1, -1, :]])
        ss_x, ss_j, _ = cbind.ss_query_batch(laps[-cfg.max_lap_stored:], tr["L"], cfg.num_ss_pts, cfg.num_ss_pts_per_lap, q)
    else:
        veh = P.barc_vehicle() if kind == "barc" else P.iac_vehicle()
        cfg = P.barc_tracking_mpc(N) if kind == "barc" else P.iac_tracking_mpc(N)
        tr = wl.synthetic_track("barc" if kind == "barc" else "putnam")
        u_lo, u_hi = Q.effective_bounds(cfg, veh)[:2]
        x, u = wl.sample_initial_states("barc" if kind == "barc" else "putnam", B, tr["L"], u_lo, u_hi, seed=seed)
        inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
    return cfg, veh, inp, ss_x, ss_j

if __name__ == "__main__":
    kind, N, B, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    cfg, veh, inp, ss_x, ss_j = build(kind, N, B, seed)
    cache = f"/tmp/ref64_{kind}_{N}_{B}_{seed}.pkl"
    if os.path.exists(cache):
        ref = pickle.load(open(cache, "rb"))
    else:  # the reference in a child process (its own environment knobs)
        code = f"""
import sys, pickle, ctypes
sys.argv = ['x', '{kind}', '{N}', '{B}', '{seed}']
sys.path.insert(0, '{ROOT}/scratch')
import r3_f32_eval as E
from oracle import cbind
cfg, veh, inp, ss_x, ss_j = E.build('{kind}', {N}, {B}, {seed})
cbind._LIB = ctypes.CDLL('/tmp/liboracle_pol.so')
o = cbind.solve_batch(cfg, veh, inp, ss_x=ss_x, ss_j=ss_j, tol=3e-14)
pickle.dump({{k: o[k] for k in ('X_optm', 'U_optm', 'dU_optm', 'status', 'iters')}}, open('{cache}', 'wb'))
"""
        env = dict(os.environ, POL_ON="1", POL_THETA="1e8", POL_MU="1e-8", POL_RD="1e-6", POL_ROUNDS="2", POL_STEPS="2", POL_FEAS="1e-9", POL_DUAL="1e-7")
        for k in ("POL_DEBUG", "POL_JOSEPH"): env.pop(k, None)
        subprocess.check_call([sys.executable, "-c", code], env=env)
        ref = pickle.load(open(cache, "rb"))
    lib = ctypes.CDLL(os.environ.get("TWIN_LIB", "/tmp/liboracle_f32.so"))
    cbind._LIB = lib
    stats = (ctypes.c_int * 8).in_dll(lib, "g_pol_stats")
    for k in range(8): stats[k] = 0
    o = cbind.solve_batch(cfg, veh, inp, ss_x=ss_x, ss_j=ss_j, tol=0.0)
    ok = (ref["status"] == 0)
    ex = np.abs((o["X_optm"] - ref["X_optm"]) / P.SCALE_X[:, None, None]).max(axis=(0, 1))
    eu = np.abs((o["U_optm"] - ref["U_optm"]) / P.SCALE_U[:, None, None]).max(axis=(0, 1))
    e = np.maximum(ex, eu)
    both = ok & (o["status"] == 0)
    print(f"{kind} N={N} B={B}: ref status {np.bincount(ref['status'], minlength=3)} iters {ref['iters'].mean():.2f} | f32 status {np.bincount(o['status'], minlength=3)} iters {o['iters'].mean():.2f} polish acc/rej/rounds/skip {list(stats)[:4]}")
    print(f"   err XU 50/90/99/99.9/max {np.quantile(e[both], [.5, .9, .99, .999, 1.0])}  frac>1e-3 {np.mean(e[both] > 1e-3):.4f}")
    acc = o["kkt"][1] < 0
    if acc.any(): print(f"   accepted {acc.sum()}: max err {e[both & acc].max():.2e}, 99.9% {np.quantile(e[both & acc], .999):.2e};  not accepted {np.sum(both & ~acc)}: max err {e[both & ~acc].max() if (both & ~acc).any() else 0:.2e}")
    if acc.any(): print("   worst accepted:", [(int(b), f"{e[b]:.1e}") for b in np.argsort(-np.where(both & acc, e, 0))[:5]])
    if os.environ.get("DUMP_ACC"): np.save(os.environ["DUMP_ACC"], acc)
    if acc.any():
        ls = o["kkt"][0]
        for lo, hi in ((0, 1e-7), (1e-7, 1e-6), (1e-6, 1e-5), (1e-5, 1e-4), (1e-4, 1e-3), (1e-3, 1e9)):
            sel = both & acc & (ls >= lo) & (ls < hi)
            if sel.any(): print(f"     last step in [{lo:g},{hi:g}): {sel.sum():5d} problems, max err {e[sel].max():.2e}")
    worst = np.argsort(-np.where(both, e, 0))[:6]
    print("   worst:", [(int(b), f"{e[b]:.1e}", int(o['iters'][b])) for b in worst])
