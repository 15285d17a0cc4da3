"""Dense restatement of the NONLINEAR-dynamics problem (test infrastructure; see oracle/__init__.py).

RacingMPC(full_dynamics = true) builds the same cost and rows as the QP but with the dynamics equalities
x_{i+1} = f_d(x_i, u_i, k_i, t_i) themselves instead of their linearisation (racing_mpc.cpp:162-166) and hands the
problem to IPOPT (:67-84); the node uses it once, for its very first solve (racing_mpc_node.cpp:299-314).  IPOPT is not
available, so "the reference's result" is pinned the solver-independent way:

  nlp_kkt_certificate   first-order optimality of a candidate point for the NLP.  The Lagrangian's stationarity involves
                        the constraint Jacobian AT the point, which is exactly the QP oracle/qp.py assembles when it
                        linearises about that point -- so the point is a KKT point of the NLP iff (a) its dynamics defect
                        vanishes and (b) it is the optimum of the QP linearised about itself.  (b) is checked with the
                        NNLS multipliers of qp.kkt_certificate, independent of any solver.
  solve_nlp_dense       an independent dense SQP (dense QPs of oracle/qp.py + the same l1-merit backtracking) from the
                        same start, to compare trajectories (the NLP is non-convex: the same start is part of "the same
                        problem").
PARITY UNPINNED like the rest of the oracle: no reference-held outputs exist for this path.
"""
from __future__ import annotations

import numpy as np

from . import dynamics as dyn
from . import qp as Q
from .params import SCALE_X, MPCConfig, Vehicle


def defect(veh: Vehicle, pr: dict, X: np.ndarray, U: np.ndarray) -> np.ndarray:
    """(x_{i+1} - f_d(x_i, u_i, k_i, t_i)) / scale_x, shape (N-1, 6)."""
    N = X.shape[1]
    kap = np.asarray(pr["curvatures"]).reshape(-1)[:N - 1]
    T = np.asarray(pr["T_ref"]).reshape(-1)
    nxt = dyn.rk4(X[:, :-1].T, U.T, kap, T, veh)
    return (X[:, 1:].T - nxt) / SCALE_X


def _at(pr: dict, X, U) -> dict:
    p = dict(pr)
    p["X_ref"], p["U_ref"] = np.asarray(X), np.asarray(U)
    return p


def nlp_kkt_certificate(cfg: MPCConfig, veh: Vehicle, pr: dict, X, U, dU, sigma=None, lam=None, eps=None) -> dict:
    """sigma = None: the boundary slack is eliminated (sigma* = the largest boundary violation of X, >= 0)."""
    qp = Q.build_qp(cfg, veh, _at(pr, X, U)) if not cfg.learning else Q.build_qp(cfg, veh, _at(pr, X, U), pr["ss_x"], pr["ss_j"])
    y = Q.pack(qp, X, U, dU, sigma=0.0 if sigma is None else max(float(sigma), 0.0), lam=lam, eps=eps)
    if sigma is None and qp.has_sigma:
        rows = (qp.C[:, qp.isig] != 0.0) & ((qp.C != 0.0).sum(axis=1) == 2)
        y[qp.isig] = max(0.0, float((qp.C[rows] @ y - qp.d[rows]).max()))
    cert = Q.kkt_certificate(qp, y)
    cert["defect"] = float(np.abs(defect(veh, pr, np.asarray(X), np.asarray(U))).max())
    # (the QP's own dynamics rows hold at its linearisation point iff the defect vanishes, so `eq` repeats `defect` in
    #  physical units plus the linear rate / initial rows)
    return cert


def merit_cost(qp: Q.DenseQP, y: np.ndarray) -> float:
    """The QP's cost at y with the boundary slack eliminated (sigma* = the largest boundary violation)."""
    y = y.copy()
    if qp.has_sigma:
        y[qp.isig] = 0.0
        rows = (qp.C[:, qp.isig] != 0.0) & ((qp.C != 0.0).sum(axis=1) == 2)
        y[qp.isig] = max(0.0, float((qp.C[rows] @ y - qp.d[rows]).max()))
    return qp.objective(y)


def solve_nlp_dense(cfg: MPCConfig, veh: Vehicle, pr: dict, max_sqp: int = 40, tol: float = 1e-9):
    """Dense SQP from (X_ref, U_ref), dU = 0: returns (X, U, dU, sigma, info)."""
    N = cfg.N
    X, U = np.array(pr["X_ref"], dtype=float), np.array(pr["U_ref"], dtype=float)
    dU, sigma, nu = np.zeros((2, N - 1)), 0.0, 1e-3
    info = {"status": 1, "sqp_iters": 0}
    prev, backoffs, move = None, 0, np.inf
    for it in range(max_sqp):
        qp = Q.build_qp(cfg, veh, _at(pr, X, U))
        y, qi = Q.solve_dense(qp)
        info["sqp_iters"] = it + 1
        if qi["status"] != 0:
            if prev is not None and backoffs < 6:   # the step outran its linearisation: half way back, linearise again
                backoffs += 1
                X, U, dU, sigma = 0.5 * (X + prev[0]), 0.5 * (U + prev[1]), 0.5 * (dU + prev[2]), 0.5 * (sigma + prev[3])
                continue
            info["status"] = 2
            break
        backoffs = 0
        o = qp.split(y)
        y0 = Q.pack(qp, X, U, dU, sigma=sigma)
        a = 1.0
        if it > 0:
            c0 = np.abs(defect(veh, pr, X, U)).sum()
            dJ = merit_cost(qp, y) - merit_cost(qp, y0)
            if c0 > 0 and dJ > 0:
                nu = max(nu, dJ / (0.9 * c0))
            D, phi0 = dJ - nu * c0, merit_cost(qp, y0) + nu * c0
            for t in range(8):
                ya = y0 + a * (y - y0)
                oa = qp.split(ya)
                if merit_cost(qp, ya) + nu * np.abs(defect(veh, pr, oa["X_optm"], oa["U_optm"])).sum() <= phi0 + 1e-4 * a * D + 1e-14 * (1 + abs(phi0)) or t == 7:
                    break
                a *= 0.5
        move = np.abs((o["X_optm"] - X) / SCALE_X[:, None]).max()   # the step proposed, whatever part of it is taken
        prev = (X, U, dU, sigma)
        X, U, dU = X + a * (o["X_optm"] - X), U + a * (o["U_optm"] - U), dU + a * (o["dU_optm"] - dU)
        sigma = sigma + a * (o.get("sigma", 0.0) - sigma)
        if move <= tol:
            info["status"] = 0
            break
    info["move"] = move
    return X, U, dU, sigma, info
