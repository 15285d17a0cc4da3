"""One car through the host entry points: lmpc_solve_host (cold) against lmpc_solve_host_warm with the optimum as the plan (accepted),
with the closed loop's shifted plan, and with noise (refused): wall-clock per call, iterations."""
import sys, time, ctypes as C, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package
from oracle import scenario as S
pkg = load_package()
for N in (20, 60):
    tr = pkg.workloads.synthetic_track("barc")
    sv = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), device=0)
    rng = np.random.default_rng(4)
    B = 64
    s0 = rng.uniform(0, tr["L"], B)
    x = np.stack([s0, rng.uniform(-0.1, 0.1, B), rng.normal(0, 0.03, B), 0.7 * S.track_lookup(tr["vel"], s0, tr["L"]), rng.normal(0, 0.02, B), rng.normal(0, 0.1, B)])
    inp = sv.prepare(tr, x.copy(), 0.025, speed_scale=0.9)
    inp["u_ic"] = torch.zeros((2, B), dtype=torch.float64, device="cuda")
    # a few closed-loop periods so that X_ref / U_ref are shifted plans
    out = sv.alloc_outputs(B)
    for _ in range(8):
        sv.solve(inp, out)
        u0 = out["U_optm"][:, 0, :].contiguous()
        xn = sv.plant_step(tr, inp["x_ic"].clone(), u0, 0.0125, 2)
        nxt = sv.shift(tr, inp, out, 0.025, speed_scale=0.9)
        nxt["x_ic"], nxt["u_ic"] = xn, u0
        inp = nxt
    sol = sv.solve(inp)
    torch.cuda.synchronize()
    h = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in inp.items()}
    so = {k: v.cpu().numpy() for k, v in sol.items() if torch.is_tensor(v)}
    P = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    lib, hd = sv.lib, sv._h
    res = {}
    for what in ("cold", "warm: the optimum as the plan", "warm: the shifted previous plan", "warm: noise"):
        t_all, it_all = [], []
        for b in range(B):
            a = dict(x_ic=P(h["x_ic"][:, b]), u_ic=P(h["u_ic"][:, b]), X_ref=P(h["X_ref"][:, :, b].T), U_ref=P(h["U_ref"][:, :, b].T), T_ref=P(h["T_ref"][:, b]),
                     bl=P(h["bound_left"][:, b]), br=P(h["bound_right"][:, b]), kap=P(h["curvatures"][:, b]), vr=P(h["vel_ref"][:, b]))
            X, U, dU = np.zeros((N, 6)), np.zeros((N - 1, 2)), np.zeros((N - 1, 2))
            st, it = C.c_int32(-1), C.c_int32(0)
            if what == "cold":
                call = lambda: lib.lmpc_solve_host(hd, ptr(a["x_ic"]), ptr(a["u_ic"]), ptr(a["X_ref"]), ptr(a["U_ref"]), ptr(a["T_ref"]), ptr(a["bl"]), ptr(a["br"]), ptr(a["kap"]), ptr(a["vr"]),
                                                   C.c_double(tr["L"]), None, None, ptr(X), ptr(U), ptr(dU), None, C.byref(st), C.byref(it))
            else:
                if "optimum" in what:
                    Xp, Up = P(so["X_optm"][:, :, b].T), P(so["U_optm"][:, :, b].T)
                elif "shifted" in what:
                    Xp, Up = a["X_ref"], a["U_ref"]
                else:
                    Xp, Up = P(rng.normal(size=(N, 6))), P(0.01 * rng.normal(size=(N - 1, 2)))
                call = lambda: lib.lmpc_solve_host_warm(hd, ptr(a["x_ic"]), ptr(a["u_ic"]), ptr(a["X_ref"]), ptr(a["U_ref"]), ptr(a["T_ref"]), ptr(a["bl"]), ptr(a["br"]), ptr(a["kap"]), ptr(a["vr"]),
                                                        C.c_double(tr["L"]), ptr(Xp), ptr(Up), ptr(X), ptr(U), ptr(dU), C.byref(st), C.byref(it))
            for r in range(6):
                t0 = time.perf_counter(); rc = call(); dt = time.perf_counter() - t0
                assert rc == 0
                if r >= 2:
                    t_all.append(dt)
            it_all.append(it.value)
            if what == "cold":
                res[b] = X.copy()
            elif st.value == 0:
                assert np.abs(X - res[b]).max() < 1e-6 * 2000
        t_all = np.array(t_all) * 1e3
        print("N = %d, one car, %s: %.3f ms per call (median; p99 %.3f), iterations mean %.2f max %d" % (N, what, np.median(t_all), np.percentile(t_all, 99), np.mean(it_all), max(it_all)), flush=True)
    sv.close()
