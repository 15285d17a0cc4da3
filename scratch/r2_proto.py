"""Round-2 prototype: the twin's interior-point iteration in numpy with a pluggable Newton solver.

Answers: on the low-speed long-horizon problems that the plain Riccati recursion loses, does the SAME iteration
converge with an exact (pivoted dense) Newton solve, and which structured factorisation keeps that accuracy?
Tracking problem only (no safe-set block).
"""
import sys
import numpy as np
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "racing-lmpc-ros2_amd"))
import workloads as wl
from oracle import params as P, qp as Q, scenario as S

NSLOT, SL_U, SL_V, SL_EY = 11, 6, 8, 10


class Stage:
    pass


def build(cfg, veh, pr):
    N = cfg.N
    A, B, g = Q.linearise(cfg, veh, pr)
    st = Stage()
    st.N = N
    t = np.asarray(pr["T_ref"]).reshape(-1)
    st.t = t
    st.Ab = np.zeros((N - 1, 8, 8)); st.Bb = np.zeros((N - 1, 8, 2)); st.gb = np.zeros((N - 1, 8))
    for i in range(N - 1):
        st.Ab[i, :6, :6] = A[i]; st.Ab[i, :6, 6:] = B[i]; st.Ab[i, 6:, 6:] = np.eye(2)
        st.Bb[i, :6] = B[i] * t[i]; st.Bb[i, 6:] = np.eye(2) * t[i]
        st.gb[i, :6] = g[i]
    qd = 2 * np.array([0, cfg.q_contour, cfg.q_heading, cfg.q_vel, cfg.q_vy, cfg.q_vyaw])
    qt = 20 * np.array([0, cfg.q_contour, cfg.q_heading, cfg.q_vel, 0, 0])
    vref = np.asarray(pr["vel_ref"]).reshape(-1)
    st.Qz = np.zeros((N, 8, 8)); st.qz = np.zeros((N, 8))
    for i in range(N):
        d = qt if i == N - 1 else qd
        st.Qz[i, :6, :6] = np.diag(d)
        st.qz[i, 3] = -d[3] * vref[i]
        if i >= 1:
            st.Qz[i, 6:, 6:] = cfg.R + cfg.R.T
    st.Sv = cfg.R_d + cfg.R_d.T
    st.has_sigma = cfg.q_boundary > 0
    st.qsig = 2 * cfg.q_boundary
    u_lo, u_hi, du_lo, du_hi = Q.effective_bounds(cfg, veh)
    marg = cfg.margin + veh.b / 2
    bl = np.asarray(pr["bound_left"]).reshape(-1); br = np.asarray(pr["bound_right"]).reshape(-1)
    st.hi = np.full((N, NSLOT), np.inf); st.lo = np.full((N, NSLOT), -np.inf)
    for i in range(N):
        if 1 <= i <= N - 2:
            st.hi[i, :6] = cfg.x_max; st.lo[i, :6] = cfg.x_min
        if i >= 1:
            st.hi[i, 6:8] = u_hi; st.lo[i, 6:8] = u_lo
        if i <= N - 2:
            st.hi[i, 8:10] = du_hi; st.lo[i, 8:10] = du_lo
        if st.has_sigma or i >= 1:
            st.hi[i, 10] = bl[i] - marg; st.lo[i, 10] = br[i] + marg
    st.z0 = np.concatenate([pr["x_ic"], pr["u_ic"]])
    return st


def slot_vals(z, v):
    N = z.shape[0]
    out = np.zeros((N, NSLOT))
    out[:, :8] = z
    out[:N - 1, 8:10] = v
    out[:, 10] = z[:, 1]
    return out


# ---------------------------------------------------------------- Newton solvers
# each: factor(st, Hz [N,8,8], Hv [N-1,2,2]) -> object with solve(qz [N,8], qv [N-1,2]) -> (dz, dv), dz_0 = 0

class DenseKKT:
    def __init__(self, st, Hz, Hv):
        N = st.N
        nz, nv = 8 * (N - 1), 2 * (N - 1)
        n = nz + nv
        me = 8 * (N - 1)
        K = np.zeros((n + me, n + me))
        for i in range(1, N):
            K[8 * (i - 1):8 * i, 8 * (i - 1):8 * i] = Hz[i]
        for i in range(N - 1):
            K[nz + 2 * i:nz + 2 * i + 2, nz + 2 * i:nz + 2 * i + 2] = Hv[i]
        for i in range(N - 1):  # dz_{i+1} - Ab dz_i - Bb dv_i = 0
            r = n + 8 * i
            K[r:r + 8, 8 * i:8 * i + 8] = np.eye(8)
            if i >= 1:
                K[r:r + 8, 8 * (i - 1):8 * i] = -st.Ab[i]
            K[r:r + 8, nz + 2 * i:nz + 2 * i + 2] = -st.Bb[i]
        K[:n, n:] = K[n:, :n].T
        import scipy.linalg as sl
        self.lu = sl.lu_factor(K)
        self.K = K
        self.n, self.nz, self.N = n, nz, N

    def solve(self, qz, qv):
        import scipy.linalg as sl
        N = self.N
        rhs = np.zeros(self.K.shape[0])
        rhs[:self.nz] = -qz[1:].reshape(-1)
        rhs[self.nz:self.n] = -qv.reshape(-1)
        sol = sl.lu_solve(self.lu, rhs)
        sol += sl.lu_solve(self.lu, rhs - self.K @ sol)
        dz = np.zeros((N, 8)); dz[1:] = sol[:self.nz].reshape(N - 1, 8)
        dv = sol[self.nz:self.n].reshape(N - 1, 2)
        return dz, dv


class Riccati:
    """the twin's recursion"""
    def __init__(self, st, Hz, Hv):
        N = st.N
        self.st = st
        self.K = np.zeros((N - 1, 2, 8)); self.Hinv = np.zeros((N - 1, 2, 2))
        Pm = Hz[N - 1].copy()
        self.Pn = np.zeros(N)
        for i in range(N - 2, -1, -1):
            Ab, Bb = st.Ab[i], st.Bb[i]
            H = Hv[i] + Bb.T @ Pm @ Bb
            G = Bb.T @ Pm @ Ab
            self.Hinv[i] = np.linalg.inv(H)
            self.K[i] = self.Hinv[i] @ G
            Pm = Hz[i] + Ab.T @ Pm @ Ab - G.T @ self.K[i]
            self.Pn[i] = np.abs(Pm).max()

    def solve(self, qz, qv):
        st, N = self.st, self.st.N
        kk = np.zeros((N - 1, 2))
        p = qz[N - 1].copy()
        for i in range(N - 2, -1, -1):
            Ab, Bb = st.Ab[i], st.Bb[i]
            hv = qv[i] + Bb.T @ p
            kk[i] = self.Hinv[i] @ hv
            p = qz[i] + Ab.T @ p - self.K[i].T @ hv
        dz = np.zeros((N, 8)); dv = np.zeros((N - 1, 2))
        for i in range(N - 1):
            dv[i] = -kk[i] - self.K[i] @ dz[i]
            dz[i + 1] = st.Ab[i] @ dz[i] + st.Bb[i] @ dv[i]
        return dz, dv


def ipm(st, Solver, max_iter=60, tol=1e-11, verbose=False, start="lq", trace=None, sig_row=True):
    N = st.N
    SR = 1.0 if (st.has_sigma and sig_row) else 0.0
    act_u = np.isfinite(st.hi); act_l = np.isfinite(st.lo)
    m = act_u.sum() + act_l.sum() + SR
    z = np.zeros((N, 8)); v = np.zeros((N - 1, 2))
    z[0] = st.z0

    def newton(Hz, Hv, qz, qv, csig=None):
        f = Solver(st, Hz, Hv)
        return f

    # ---- start point
    Hz0 = st.Qz.copy(); Hv0 = np.repeat(st.Sv[None], N - 1, 0)
    f = Solver(st, Hz0, Hv0)
    if start == "lq":
        if hasattr(f, "K"):
            Kfb = f.K if isinstance(f.K, np.ndarray) and f.K.ndim == 3 else None
        else:
            Kfb = None
        if Kfb is None:
            Kfb = Riccati(st, Hz0, Hv0).K
        for i in range(N - 1):
            v[i] = -Kfb[i] @ z[i]
            z[i + 1] = st.Ab[i] @ z[i] + st.Bb[i] @ v[i] + st.gb[i]
        gz = np.einsum("irc,ic->ir", st.Qz, z) + st.qz
        gv = v @ st.Sv.T
        dz, dv = f.solve(gz, gv)
        z += dz; v += dv
    sv = slot_vals(z, v)
    rng = np.where(act_u & act_l, st.hi - st.lo, 1.0)
    rng = np.maximum(np.nan_to_num(rng, nan=1.0, posinf=1.0), 1e-3)
    tu = np.where(act_u, np.maximum(st.hi - sv, 0.5 * rng), 1.0); tl = np.where(act_l, np.maximum(sv - st.lo, 0.5 * rng), 1.0)
    lu = np.where(act_u, 0.1 / tu, 0.0); ll = np.where(act_l, 0.1 / tl, 0.0)
    sigma, ts, lams = 0.0, 0.1, SR
    status = 1
    hiF = np.where(act_u, st.hi, 0.0); loF = np.where(act_l, st.lo, 0.0)
    for it in range(max_iter + 1):
        sv = slot_vals(z, v)
        sg = np.zeros((N, NSLOT)); sg[:, 10] = sigma if st.has_sigma else 0.0
        rdu = np.where(act_u, sv - sg + tu - hiF, 0.0); rdl = np.where(act_l, -sv - sg + tl + loF, 0.0)
        thu = lu / tu; thl = ll / tl
        mu = ((lu * tu).sum() + (ll * tl).sum() + ts * lams * SR) / m
        rd = max(np.abs(rdu).max(), np.abs(rdl).max(), abs(-sigma + ts) if SR else 0.0)
        # dynamics residual
        dyn = max(np.abs(z[i + 1] - st.Ab[i] @ z[i] - st.Bb[i] @ v[i] - st.gb[i]).max() for i in range(N - 1))
        if verbose:
            print(f"  it {it}: mu {mu:.2e} rd {rd:.2e} dyn {dyn:.1e} sigma {sigma:.2e} |z| {np.abs(z).max():.2e}")
        if not np.isfinite(mu):
            status = 2; break
        if mu <= tol and rd <= 1e-9:
            status = 0; break
        if it == max_iter:
            break
        th = thu + thl
        Hz = st.Qz.copy(); Hv = np.repeat(st.Sv[None], N - 1, 0).copy()
        for k in range(8):
            Hz[:, k, k] += th[:, k]
        Hz[:, 1, 1] += th[:, 10]
        Hv[:, 0, 0] += th[:N - 1, 8]; Hv[:, 1, 1] += th[:N - 1, 9]
        csig = (thl[:, 10] - thu[:, 10]) if st.has_sigma else np.zeros(N)
        hsig = st.qsig + th[:, 10].sum() + SR * lams / ts if st.has_sigma else 1.0
        f = Solver(st, Hz, Hv)
        if trace is not None:
            trace.append((Hz, Hv, f))
        if st.has_sigma:
            cz = np.zeros((N, 8)); cz[1:, 1] = csig[1:]
            ez, ev = f.solve(cz, np.zeros((N - 1, 2)))
            ce = (csig[1:] * ez[1:, 1]).sum()
        gz0 = np.einsum("irc,ic->ir", st.Qz, z) + st.qz
        gv0 = v @ st.Sv.T
        sigc = 0.0; pu = np.zeros_like(tu); pl = np.zeros_like(tl); dts = dlams = 0.0
        for pas in range(2):
            smu = sigc * mu if pas else 0.0
            cu = np.where(act_u, thu * rdu + (smu - pas * pu) / tu, 0.0)
            cl = np.where(act_l, thl * rdl + (smu - pas * pl) / tl, 0.0)
            dl = cu - cl
            qz = gz0.copy(); qv = gv0.copy()
            qz[:, :8] += dl[:, :8]; qz[:, 1] += dl[:, 10]; qv += dl[:N - 1, 8:10]
            dz, dv = f.solve(qz, qv)
            dsig = 0.0
            if st.has_sigma:
                cfs = SR * (lams / ts * (-sigma + ts) + (smu - pas * dts * dlams) / ts)
                qsg = st.qsig * sigma - (cu[:, 10] + cl[:, 10]).sum() - cfs
                dsig = -(qsg + (csig[1:] * dz[1:, 1]).sum()) / (hsig + ce)
                dz = dz + dsig * ez; dv = dv + dsig * ev
            dsv = slot_vals(dz, dv)
            dsg = np.zeros((N, NSLOT)); dsg[:, 10] = dsig
            dtu = np.where(act_u, -rdu - (dsv - dsg), 0.0); dtl = np.where(act_l, -rdl - (-dsv - dsg), 0.0)
            dlu = np.where(act_u, -lu + (smu - pas * pu) / tu - thu * dtu, 0.0)
            dll = np.where(act_l, -ll + (smu - pas * pl) / tl - thl * dtl, 0.0)
            amax = 1.0
            for t_, d_ in ((tu, dtu), (tl, dtl), (lu, dlu), (ll, dll)):
                neg = d_ < 0
                if neg.any():
                    amax = min(amax, float((-t_[neg] / d_[neg]).min()))
            if SR:
                dts = -(-sigma + ts) + dsig
                dlams = -lams + cfs - lams / ts * (-sigma + ts) - lams / ts * dts
                if dts < 0: amax = min(amax, -ts / dts)
                if dlams < 0: amax = min(amax, -lams / dlams)
            if pas == 0:
                s = ((tu + amax * dtu) * (lu + amax * dlu)).sum() + ((tl + amax * dtl) * (ll + amax * dll)).sum()
                if SR: s += (ts + amax * dts) * (lams + amax * dlams)
                sigc = ((s / m) / mu) ** 3
                pu = dtu * dlu; pl = dtl * dll
        alpha = min(1.0, 0.995 * amax)
        if not np.isfinite(alpha) or not np.isfinite(dz).all():
            status = 2; break
        z[1:] += alpha * dz[1:]; v += alpha * dv
        tu += alpha * dtu; tl += alpha * dtl; lu += alpha * dlu; ll += alpha * dll
        if st.has_sigma:
            sigma += alpha * dsig; ts += alpha * dts * SR; lams += alpha * dlams * SR
    return {"z": z, "v": v, "sigma": sigma, "status": status, "iters": it, "mu": mu, "rd": rd}


def setup(N, B=256, seed=0):
    veh, cfg = P.barc_vehicle(), P.barc_tracking_mpc(N)
    tr = wl.synthetic_track("barc")
    x, u = wl.sample_initial_states("barc", B, tr["L"], [-0.015, -0.314159], [0.015, 0.314159], seed=seed)
    inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
    return cfg, veh, inp, x


def dense_ref(cfg, veh, pr):
    qp = Q.build_qp(cfg, veh, pr)
    y, info = Q.solve_dense(qp)
    return qp.split(y), info


def err_vs(res, ex):
    X = res["z"][:, :6].T; U = res["z"][1:, 6:].T; dU = res["v"].T
    ex_ = np.abs(X - ex["X_optm"]) / P.SCALE_X[:, None]
    eu = np.abs(U - ex["U_optm"]) / P.SCALE_U[:, None]
    return ex_.max(), eu.max()


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    bs = [int(a) for a in sys.argv[2:]] or [17, 0, 3]
    cfg, veh, inp, x = setup(N)
    for b in bs:
        pr = S.problem(inp, b)
        st = build(cfg, veh, pr)
        ex, info = dense_ref(cfg, veh, pr)
        print(f"b={b} vx0 {x[b,3]:.2f} dense status {info['status']} polished {info.get('polished')}")
        for name, Sv in (("exact", DenseKKT), ("riccati", Riccati)):
            r = ipm(st, Sv, verbose="-v" in sys.argv)
            print(f"  {name}: status {r['status']} it {r['iters']} mu {r['mu']:.1e} rd {r['rd']:.1e} err X,U {err_vs(r, ex)}")
