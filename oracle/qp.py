"""Dense restatement of the reference's per-step QP + reference solve + KKT certificate.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: the reference
hands this QP to OSQP through CasADi's Opti("conic") (racing_mpc.cpp:42,86-103,
344); neither is available, and the reference holds no golden solutions.  The
QP has a unique optimum in (X, U, dU, slack) because R_d > 0, R > 0 and the
states follow from the equality constraints, so "the reference's result" is
that optimum up to OSQP's own 1e-3 tolerance; the oracle computes it to ~1e-12
and certifies it with solver-independent KKT residuals.

Variable order (as RacingMPC declares them, racing_mpc.cpp:43-45,533,484,495):
    y = [ X(:) (6N, column-major) | U(:) (2(N-1)) | dU(:) (2(N-1)) | sigma (1 iff q_boundary>0)
          | lambda (S) | eps (6) ]     (last two only when learning)
in PHYSICAL units (the reference optimises X/scale_x etc.; scaling does not
move the optimum).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
from scipy.linalg import lu_factor, lu_solve

from . import dynamics as dyn
from .params import SCALE_U, SCALE_X, MPCConfig, Vehicle

NX, NU = 6, 2
_TRACE = bool(__import__('os').environ.get('DENSE_TRACE'))


@dataclass
class DenseQP:
    H: np.ndarray
    h: np.ndarray
    c0: float
    A: np.ndarray
    b: np.ndarray
    C: np.ndarray  # C y <= d, one-sided rows, infinite bounds dropped
    d: np.ndarray
    N: int
    has_sigma: bool
    S: int  # number of safe-set points (0 for tracking)

    # index helpers
    def ix(self, i, k):
        return NX * i + k

    def iu(self, i, k):
        return NX * self.N + NU * i + k

    def idu(self, i, k):
        return NX * self.N + NU * (self.N - 1) + NU * i + k

    @property
    def isig(self):
        return NX * self.N + 2 * NU * (self.N - 1)

    @property
    def ilam(self):
        return self.isig + (1 if self.has_sigma else 0)

    @property
    def ieps(self):
        return self.ilam + self.S

    @property
    def n(self):
        return self.H.shape[0]

    def split(self, y):
        N = self.N
        X = y[: NX * N].reshape(N, NX).T.copy()
        U = y[NX * N: NX * N + NU * (N - 1)].reshape(N - 1, NU).T.copy()
        dU = y[NX * N + NU * (N - 1): NX * N + 2 * NU * (N - 1)].reshape(N - 1, NU).T.copy()
        out = {"X_optm": X, "U_optm": U, "dU_optm": dU}
        if self.has_sigma:
            out["sigma"] = float(y[self.isig])
        if self.S:
            out["convex_combi_optm"] = y[self.ilam: self.ilam + self.S].copy()
            out["eps"] = y[self.ieps: self.ieps + NX].copy()
        return out

    def objective(self, y):
        return 0.5 * y @ self.H @ y + self.h @ y + self.c0


def linearise(cfg: MPCConfig, veh: Vehicle, inp: dict):
    """(A_i, B_i, g_i) for i = 0..N-2 about (X_ref, U_ref) -- racing_mpc.cpp:169-186."""
    N = cfg.N
    Xr = np.asarray(inp["X_ref"], dtype=np.float64)
    Ur = np.asarray(inp["U_ref"], dtype=np.float64)
    Tr = np.asarray(inp["T_ref"], dtype=np.float64).reshape(-1)
    kap = np.asarray(inp["curvatures"], dtype=np.float64).reshape(-1)
    A, B, g = dyn.rk4_jacobian_cs(Xr[:, : N - 1].T, Ur.T, kap[: N - 1], Tr, veh)
    return A, B, g


def effective_bounds(cfg: MPCConfig, veh: Vehicle):
    """Intersect the MPC's u box (racing_mpc.cpp:148) with the model's actuator box
    (single_track_planar_model.cpp:114,120); rate box from :146-151."""
    u_lo = np.maximum(cfg.u_min, np.array([veh.Fb_max / 1000.0, -veh.max_steer]))
    u_hi = np.minimum(cfg.u_max, np.array([veh.Fd_max / 1000.0, veh.max_steer]))
    du_lo = np.array([veh.Fb_max / 1000.0 / veh.Tb, -veh.max_steer_rate])
    du_hi = np.array([veh.Fd_max / 1000.0 / veh.Td, veh.max_steer_rate])
    return u_lo, u_hi, du_lo, du_hi


def build_qp(cfg: MPCConfig, veh: Vehicle, inp: dict, ss_x=None, ss_j=None, lin=None) -> DenseQP:
    """Assemble the QP exactly as RacingMPC::RacingMPC builds it (racing_mpc.cpp:106-201).

    lin = (A [N-1,6,6], B [N-1,6,2], g [N-1,6]): stage models to use INSTEAD of the linearisation about (X_ref, U_ref) -- the model
    with the error-dynamics regression's correction added (safe_set.cpp:182-245), which has no caller upstream and therefore no place
    in the reference's problem definition; the tests that compare on regressed models pass the product's corrected (A, B, g) here."""
    N = cfg.N
    learning = cfg.learning
    S = 0
    if learning:
        ss_x = np.asarray(ss_x, dtype=np.float64)
        ss_j = np.asarray(ss_j, dtype=np.float64).reshape(-1)
        S = ss_x.shape[1]
    has_sigma = cfg.q_boundary > 0.0
    n = NX * N + 2 * NU * (N - 1) + (1 if has_sigma else 0) + ((S + NX) if learning else 0)
    qp = DenseQP(np.zeros((n, n)), np.zeros(n), 0.0, None, None, None, None, N, has_sigma, S)
    H, h = qp.H, qp.h
    x_ic = np.asarray(inp["x_ic"], dtype=np.float64).reshape(-1)
    u_ic = np.asarray(inp["u_ic"], dtype=np.float64).reshape(-1)
    Tr = np.asarray(inp["T_ref"], dtype=np.float64).reshape(-1)
    bl = np.asarray(inp["bound_left"], dtype=np.float64).reshape(-1)
    br = np.asarray(inp["bound_right"], dtype=np.float64).reshape(-1)
    vref = np.asarray(inp["vel_ref"], dtype=np.float64).reshape(-1)
    Ad, Bd, gd = linearise(cfg, veh, inp) if lin is None else (np.asarray(a, dtype=np.float64) for a in lin)

    R2 = cfg.R + cfg.R.T
    Rd2 = cfg.R_d + cfg.R_d.T
    # --- cost ---
    if not learning:
        # build_tracking_cost, racing_mpc.cpp:442-477
        qd = np.array([0.0, cfg.q_contour, cfg.q_heading, cfg.q_vel, cfg.q_vy, cfg.q_vyaw])
        for i in range(N - 1):
            for k in range(NX):
                H[qp.ix(i, k), qp.ix(i, k)] += 2 * qd[k]
            h[qp.ix(i, 3)] += -2 * cfg.q_vel * vref[i]
            qp.c0 += cfg.q_vel * vref[i] ** 2
        qt = 10.0 * np.array([0.0, cfg.q_contour, cfg.q_heading, cfg.q_vel, 0.0, 0.0])
        for k in range(NX):
            H[qp.ix(N - 1, k), qp.ix(N - 1, k)] += 2 * qt[k]
        h[qp.ix(N - 1, 3)] += -2 * 10.0 * cfg.q_vel * vref[N - 1]
        qp.c0 += 10.0 * cfg.q_vel * vref[N - 1] ** 2
    # u and dU effort: every column once (index -1 quirk, :451-452 / :508-509)
    for i in range(N - 1):
        for a in range(NU):
            for c in range(NU):
                H[qp.iu(i, a), qp.iu(i, c)] += R2[a, c]
                H[qp.idu(i, a), qp.idu(i, c)] += Rd2[a, c]
    if has_sigma:
        H[qp.isig, qp.isig] += 2 * cfg.q_boundary  # :539
    if learning:
        # build_lmpc_cost, :479-522
        # all-zero convex_hull_slack: no slack variable, x_T = SS lambda (:500-502) -- eps is pinned to zero below
        for k in range(NX):
            H[qp.ieps + k, qp.ieps + k] += 2 * cfg.convex_hull_slack[k]
        h[qp.ilam: qp.ilam + S] += ss_j

    # --- equalities ---
    rowsA, rhsb = [], []

    def eq(coefs, rhs):
        r = np.zeros(n)
        for j, c in coefs:
            r[j] += c
        rowsA.append(r)
        rhsb.append(rhs)

    for i in range(N - 1):
        # x_{i+1} = A x_i + B u_i + g   (:186)
        for k in range(NX):
            co = [(qp.ix(i + 1, k), 1.0)]
            co += [(qp.ix(i, c), -Ad[i, k, c]) for c in range(NX)]
            co += [(qp.iu(i, c), -Bd[i, k, c]) for c in range(NU)]
            eq(co, gd[i, k])
        # u_{i-1} + dU_i t_i = u_i   (:190-196)
        for k in range(NU):
            co = [(qp.iu(i, k), 1.0), (qp.idu(i, k), -Tr[i])]
            if i == 0:
                eq(co, u_ic[k])
            else:
                co.append((qp.iu(i - 1, k), -1.0))
                eq(co, 0.0)
    for k in range(NX):  # x_0 = x_ic (:200-201)
        eq([(qp.ix(0, k), 1.0)], x_ic[k])
    if learning:
        eq([(qp.ilam + j, 1.0) for j in range(S)], 1.0)  # :491
        for k in range(NX):  # x_{N-1} = SS lambda + eps (:496)
            co = [(qp.ix(N - 1, k), 1.0), (qp.ieps + k, -1.0)]
            co += [(qp.ilam + j, -ss_x[k, j]) for j in range(S)]
            eq(co, 0.0)
        if not np.any(cfg.convex_hull_slack > 0):
            for k in range(NX):
                eq([(qp.ieps + k, 1.0)], 0.0)
    qp.A = np.array(rowsA)
    qp.b = np.array(rhsb)

    # --- inequalities (one-sided rows C y <= d) ---
    rowsC, rhsd = [], []

    def le(coefs, rhs):
        if not np.isfinite(rhs):
            return
        r = np.zeros(n)
        for j, c in coefs:
            r[j] += c
        rowsC.append(r)
        rhsd.append(rhs)

    u_lo, u_hi, du_lo, du_hi = effective_bounds(cfg, veh)
    marg = cfg.margin + veh.b / 2.0  # :531
    for i in range(N):  # build_boundary_constraint, all N knots (:524-543)
        sg = [(qp.isig, -1.0)] if has_sigma else []
        le([(qp.ix(i, 1), 1.0)] + sg, bl[i] - marg)
        le([(qp.ix(i, 1), -1.0)] + sg, -(br[i] + marg))
    if has_sigma:
        le([(qp.isig, -1.0)], 0.0)
    for i in range(N - 1):  # stage loop :126-197 -- not applied at the terminal knot
        for k in range(NX):
            le([(qp.ix(i, k), 1.0)], cfg.x_max[k])
            le([(qp.ix(i, k), -1.0)], -cfg.x_min[k])
        for k in range(NU):
            le([(qp.iu(i, k), 1.0)], u_hi[k])
            le([(qp.iu(i, k), -1.0)], -u_lo[k])
            le([(qp.idu(i, k), 1.0)], du_hi[k])
            le([(qp.idu(i, k), -1.0)], -du_lo[k])
    if learning:
        for j in range(S):
            le([(qp.ilam + j, -1.0)], 0.0)  # :490
    qp.C = np.array(rowsC)
    qp.d = np.array(rhsd)
    return qp


def variable_scales(qp: DenseQP) -> np.ndarray:
    """The reference optimises X / scale_x, U / scale_u, dU / scale_u (racing_mpc.cpp:36-37,127-129,141): the diagonal
    D with y = D y_scaled.  sigma is declared unscaled (:533) and is a length in e_y's units, the simplex weights are O(1)
    (:484-491) and the hull residual is a state (:495)."""
    N = qp.N
    D = np.ones(qp.n)
    D[: NX * N] = np.tile(SCALE_X, N)
    D[NX * N: NX * N + NU * (N - 1)] = np.tile(SCALE_U, N - 1)
    D[NX * N + NU * (N - 1): NX * N + 2 * NU * (N - 1)] = np.tile(SCALE_U, N - 1)
    if qp.S:
        D[qp.ieps: qp.ieps + NX] = SCALE_X
    return D


def solve_dense(qp: DenseQP, tol: float = 1e-9, mu_tol: float = 1e-15, max_iter: int = 80, scaled: bool = True):
    """Mehrotra predictor-corrector IPM on the dense KKT system + active-set polish.

    Returns (y, info), y and the multipliers info["lam"], info["pi"] in PHYSICAL units (multipliers of the rows of
    qp.A / qp.C as built).  The polish re-solves the equality-constrained QP on the detected active set (what OSQP's
    polish=true does, racing_mpc.cpp:92) so the returned point is the optimum to rounding.

    Round 5: the iteration runs on the problem the reference hands its solver -- in the SCALED
    variables y_s = D^-1 y (variable_scales) -- with every equality and inequality row brought to unit infinity norm.
    In physical units the IAC problem (R = diag(1e-5, 1), abscissae of 2000 m next to yaw rates of 0.01) has a KKT matrix
    whose entries span 13 decades and the iteration stalled at mu ~ 0.05 on 9 % of the cold-start distribution; scaled,
    it solves the whole distribution.  `scaled=False` is the round-1..4 behaviour, kept for the A/B in the tests.
    """
    n, me, mi = qp.H.shape[0], qp.A.shape[0], qp.C.shape[0]
    D = variable_scales(qp) if scaled else np.ones(n)
    H = D[:, None] * qp.H * D[None, :]
    h = D * qp.h
    A = qp.A * D[None, :]
    C = qp.C * D[None, :]
    ra = np.abs(A).max(axis=1) if scaled else np.ones(me)
    rc = np.abs(C).max(axis=1) if scaled else np.ones(mi)
    ra[ra == 0.0] = 1.0
    rc[rc == 0.0] = 1.0
    A, b = A / ra[:, None], qp.b / ra
    C, d = C / rc[:, None], qp.d / rc

    def objective(ys):
        return 0.5 * ys @ H @ ys + h @ ys + qp.c0

    # least-squares point on the equalities as a start
    y = np.linalg.lstsq(A, b, rcond=None)[0]
    t = np.maximum(d - C @ y, 1.0)
    lam = np.ones(mi)
    pi = np.zeros(me)
    info = {"status": 1, "iters": max_iter}
    loose = None
    for it in range(max_iter):
        r_g = H @ y + h + A.T @ pi + C.T @ lam
        r_b = A @ y - b
        r_d = C @ y - d + t
        mu = float(lam @ t) / mi
        # stationarity relative to the size of its terms (the IAC gradient is ~5e3 in the scaled variables: an absolute 1e-9
        # is below the rounding of the sum, and an iteration pushed past mu ~ 1e-20 to get there falls apart)
        g_scale = max(1.0, np.abs(H @ y + h).max(), np.abs(A.T @ pi).max(), np.abs(C.T @ lam).max())
        res = max(np.abs(r_g).max() / g_scale, np.abs(r_b).max(), np.abs(r_d).max())
        if _TRACE:
            print(f"  it {it} mu {mu:.3e} r_g {np.abs(r_g).max():.3e} r_b {np.abs(r_b).max():.3e} r_d {np.abs(r_d).max():.3e}")
        if res < tol and mu < mu_tol:
            info = {"status": 0, "iters": it}
            break
        # Past mu ~ 1e-12 the condensed matrix H + C' diag(lam / t) C carries weights of 1e12 and more, and on some problems
        # the Newton directions lose the digits the last two decades of mu need: the residual climbs again and the iteration
        # falls apart.  The active set is decided long before that, so the latest iterate that was converged LOOSELY (mu <=
        # 1e-11, residuals <= 1e-7) is kept, and if the strict test is never met the polish below starts from it: an accepted
        # polish is a verified KKT point -- the optimum -- whatever the accuracy of the iterate that proposed its active set.
        if res < 1e-7 and mu < 1e-11:
            loose = (y.copy(), pi.copy(), t.copy(), lam.copy(), it, res)
        elif loose is not None and res > 1e3 * loose[5]:
            break
        th = lam / t
        K = np.block([[H + C.T @ (th[:, None] * C), A.T], [A, np.zeros((me, me))]])

        lu = lu_factor(K, check_finite=False)

        def newton(rm):
            rhs1 = -r_g + C.T @ (rm / t - th * r_d)
            rhs = np.concatenate([rhs1, -r_b])
            sol = lu_solve(lu, rhs, check_finite=False)
            sol = sol + lu_solve(lu, rhs - K @ sol, check_finite=False)   # one step of iterative refinement
            dy, dpi = sol[:n], sol[n:]
            dt = -r_d - C @ dy
            dlam = -(rm + lam * dt) / t
            return dy, dpi, dt, dlam

        def steplen(dt, dlam):
            a = 1.0
            neg = dt < 0
            if neg.any():
                a = min(a, float((-t[neg] / dt[neg]).min()))
            neg = dlam < 0
            if neg.any():
                a = min(a, float((-lam[neg] / dlam[neg]).min()))
            return a

        dy, dpi, dt, dlam = newton(lam * t)
        a_aff = steplen(dt, dlam)
        mu_aff = float((lam + a_aff * dlam) @ (t + a_aff * dt)) / mi
        sig = (mu_aff / mu) ** 3
        dy, dpi, dt, dlam = newton(lam * t + dt * dlam - sig * mu)
        a = min(1.0, 0.995 * steplen(dt, dlam))
        y = y + a * dy
        pi = pi + a * dpi
        t = t + a * dt
        lam = lam + a * dlam
    from_loose = info["status"] != 0 and loose is not None
    if from_loose:
        y, pi, t, lam = loose[:4]
        info["iters"] = loose[4]
    info["mu"] = float(lam @ t) / mi
    # ---- polish: the equality-constrained QP on the active set, repaired like a primal-dual active-set method ----
    # Start from the rows the interior point holds (lam > slack).  Solve; a held row with a negative multiplier is released
    # (the most negative ones first), a row the new point violates is held; repeat.  On a strictly complementary problem the
    # first guess is right; on a degenerate one (a weakly active row can sit on either side of the guess: the point is the
    # same) a round or two settle it.  Accepted = primal feasible to 1e-10, multipliers >= -1e-9, objective within 1e-6 of the
    # interior point's: a verified KKT point, i.e. THE optimum (the QP is strictly convex in the variables that matter).
    # Until round 5 there was no repair, and the ~1 % of problems whose first guess failed kept the interior point's answer,
    # which on a degenerate problem at the floor of the condensed system is 1e-5 off in dU (found by holding the twin to it).
    slack = d - C @ y
    act = lam > np.maximum(slack, 0.0)
    reg = 1e-13
    info["polished"] = False
    obj_ipm = objective(y)
    for rnd in range(8):
        Ca = C[act]
        na = Ca.shape[0]
        K = np.block([[H + reg * np.eye(n), A.T, Ca.T],
                      [A, -reg * np.eye(me), np.zeros((me, na))],
                      [Ca, np.zeros((na, me)), -reg * np.eye(na)]])
        rhs = np.concatenate([-h, b, d[act]])
        try:
            luK = lu_factor(K, check_finite=False)
            sol = lu_solve(luK, rhs, check_finite=False)
            # iterative refinement against the unregularised system
            K0 = np.block([[H, A.T, Ca.T], [A, np.zeros((me, me + na))], [Ca, np.zeros((na, me + na))]])
            for _ in range(3):
                sol = sol + lu_solve(luK, rhs - K0 @ sol, check_finite=False)
        except (np.linalg.LinAlgError, ValueError):
            break
        yp = sol[:n]
        lam_a = sol[n + me:]
        sl = d - C @ yp
        viol = (~act) & (sl < -1e-10)
        neg = np.zeros(mi, dtype=bool)
        if na and lam_a.min() < -1e-9:
            idx = np.nonzero(act)[0]
            neg[idx[lam_a <= 0.5 * lam_a.min()]] = True
        if not viol.any() and not neg.any():
            if abs(objective(yp) - obj_ipm) < 1e-6 * (1 + abs(obj_ipm)):
                y = yp
                info["polished"] = True
                info["polish_rounds"] = rnd + 1
                lam = np.zeros(mi)
                lam[act] = lam_a
                pi = sol[n: n + me]
                if from_loose:
                    info["status"] = 0
                    info["loose"] = True
            break
        act = (act & ~neg) | (viol if not neg.any() else False)
    # back to the rows and variables as built: C_s = R_c^-1 C D  =>  C' lam = D^-1 C_s' R_c lam_s ... lam = lam_s / r_c
    info["lam"] = lam / rc
    info["pi"] = pi / ra
    return D * y, info


def kkt_certificate(qp: DenseQP, y: np.ndarray, act_tol: float = 1e-5) -> dict:
    """Solver-independent optimality certificate for a primal point y.

    Finds multipliers (pi free, lam >= 0 supported on rows with slack <= act_tol)
    minimising the stationarity residual by NNLS and reports
      stat    ||H y + h + A' pi + C' lam||_inf
      eq      ||A y - b||_inf
      ineq    max(C y - d)_+
      comp    max lam_j * slack_j
    """
    from scipy.optimize import nnls

    grad = qp.H @ y + qp.h
    slack = qp.d - qp.C @ y
    act = slack <= act_tol
    Ca = qp.C[act]
    M = np.hstack([qp.A.T, -qp.A.T, Ca.T])
    mu, _ = nnls(M, -grad, maxiter=50 * M.shape[1])
    res = M @ mu + grad
    lam = mu[2 * qp.A.shape[0]:]
    return {
        "stat": float(np.abs(res).max()),
        "eq": float(np.abs(qp.A @ y - qp.b).max()),
        "ineq": float(np.maximum(-slack, 0.0).max()),
        "comp": float((lam * np.maximum(slack[act], 0.0)).max()) if lam.size else 0.0,
        "n_active": int(act.sum()),
    }


def strict_complementarity(qp: DenseQP, y: np.ndarray, lam: np.ndarray) -> float:
    """min over the inequality rows of max(lam_j, slack_j) at the dense optimum (physical units): the margin by
    which strict complementarity holds.  An interior-point iterate with complementarity mu sits ~ mu / margin from
    the optimum in the rows that attain the minimum (and ~ sqrt(mu) when the margin is zero), so this number -- not
    the solver under test -- decides which problems can be held to the contract tolerance; tests label a problem
    DEGENERATE when it is below DEGENERATE_MARGIN.

    The row  -sigma <= 0  is left out when it is decoupled: sigma* = 0 with every boundary row inactive (multiplier
    0).  It is then degenerate by construction (sigma* = 0 and, by stationarity 2 q sigma = sum of the boundary
    multipliers + its own, multiplier 0) in every problem that stays inside the track, but sigma enters nothing
    else, so X, U, dU do not depend on how that row is resolved."""
    slack = qp.d - qp.C @ y
    keep = np.ones(qp.C.shape[0], dtype=bool)
    if qp.has_sigma:
        nnz = (qp.C != 0.0).sum(axis=1)
        on_sig = qp.C[:, qp.isig] != 0.0
        sig_row = on_sig & (nnz == 1)
        boundary = on_sig & (nnz == 2)
        if abs(y[qp.isig]) <= 1e-12 and not (lam[boundary] > 1e-12).any():
            keep &= ~sig_row
    return float(np.maximum(lam, slack)[keep].min())


DEGENERATE_MARGIN = 1e-4


def pack(qp: DenseQP, X, U, dU, sigma=None, lam=None, eps=None) -> np.ndarray:
    """Inverse of DenseQP.split (sigma/eps recomputed optimally if omitted)."""
    N = qp.N
    y = np.zeros(qp.n)
    y[: NX * N] = np.asarray(X).T.reshape(-1)
    y[NX * N: NX * N + NU * (N - 1)] = np.asarray(U).T.reshape(-1)
    y[NX * N + NU * (N - 1): NX * N + 2 * NU * (N - 1)] = np.asarray(dU).T.reshape(-1)
    if qp.has_sigma:
        y[qp.isig] = 0.0 if sigma is None else sigma
    if qp.S:
        y[qp.ilam: qp.ilam + qp.S] = lam
        y[qp.ieps: qp.ieps + NX] = eps
    return y
