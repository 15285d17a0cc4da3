"""CPU restatement (test infrastructure; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it) of the reference's error-dynamics regression -- BASELINE config 5:

  SSTrajectory::query(RegQuery)      src/vehicle_dynamics_models/racing_trajectory/src/safe_set.cpp:56-114
  SafeSetManager::query(RegQuery)    safe_set.cpp:182-245
  SSTrajectory::process_lap_data     safe_set.cpp:116-137   (dt_i = t_i - t_{i+1}, negative as written, :130-135)

Parity unpinned: the reference has no caller and no test for this query.  Two of its expressions do not type-check as
written -- the nominal model `query.f` is handed the in_state rows of the lap instead of the six-component state
(:213-216), and the residual is taken on the in_state rows for every regressed output (:227) so that `R(Slice(0, ns))`
would read the first column only.  The reading restated here (and in csrc/lmpc_reg_kernel.hip):
  * the nominal RK4 step is evaluated on the full recorded state, x_pred = f_d(x_j, u_j, k_j, dt_j);
  * for the regressed row r the residual is that row's: y_r = x_{j+1}[r] - x_pred[r];
  * one (in_state, in_ctrl) feature list shared by all regressed rows (the reference allows one list per row);
  * everything else as written: the last sample of a lap is dropped, candidates with d < dist_max, weights
    K = 0.75/h (1 - (d/h)^2)^2 (:224-225), M = [xs' us' 1], Q = M'KM + 1e-3 I, R = Q^-1 b,
    A[r, in_state] += R[0:ns], B[r, in_ctrl] += R[ns:-1], C[r] += R[-1] (:235-242); a query with no candidate leaves
    (A, B, C) untouched (:207-210).
Two signs, one switch.  AS WRITTEN upstream the nominal step is taken with dt_j = t_j - t_{j+1} < 0 (process_lap_data
stores the differences that way round, :130-135) -- the model is integrated backwards in time, so the "residual" is about
twice the true step -- and the right-hand side is b = -M'K y (:229-231): the correction points away from the data, which
is consistent with the query never being called.  `as_written=True` restates exactly that.  The default is the
regression that does what its name says: dt_j = t_{j+1} - t_j, b = +M'K y, so that A x + B u + C moves TOWARDS the
recorded successor states (tests/test_regression_oracle.py: a plant with a perturbed parameter is predicted better
after the correction).  The product applies the default in front of a solve (include/lmpc_hip.h).
"""
from __future__ import annotations

import numpy as np

from .dynamics import rk4
from .params import Vehicle


def lap_residuals(veh: Vehicle, x: np.ndarray, u: np.ndarray, k: np.ndarray, t: np.ndarray, as_written: bool = False) -> np.ndarray:
    """x [n, 6], u [n, 2], k [n], t [n] of one lap -> one-step residuals [n - 1, 6] of the nominal model."""
    n = x.shape[0]
    y = np.zeros((n - 1, 6))
    for j in range(n - 1):
        dt = float(t[j] - t[j + 1]) if as_written else float(t[j + 1] - t[j])
        y[j] = x[j + 1] - rk4(x[j], u[j], float(k[j]), dt, veh)
    return y


def regress(veh: Vehicle, laps: list, in_state, in_ctrl, out_rows, dist_max: float, q_x: np.ndarray, q_u: np.ndarray,
            A: np.ndarray, B: np.ndarray, C: np.ndarray, as_written: bool = False):
    """laps: list of (x [n,6], u [n,2], k [n], t [n]); linearisation point (q_x [6], q_u [2]); returns updated copies."""
    A, B, C = A.copy(), B.copy(), C.copy()
    in_state, in_ctrl = list(in_state), list(in_ctrl)
    q = np.concatenate([q_x[in_state], q_u[in_ctrl]])
    Z, Y = [], []
    for (x, u, k, t) in laps:
        Z.append(np.concatenate([x[:-1][:, in_state], u[:-1][:, in_ctrl]], axis=1))
        Y.append(lap_residuals(veh, x, u, k, t, as_written))
    Z, Y = np.concatenate(Z), np.concatenate(Y)
    d = np.sqrt(((Z - q) ** 2).sum(axis=1))
    m = d < dist_max
    if not m.any():
        return A, B, C
    Z, Y, d = Z[m], Y[m], d[m]
    K = 0.75 / dist_max * (1 - (d / dist_max) ** 2) ** 2
    M = np.concatenate([Z, np.ones((Z.shape[0], 1))], axis=1)
    Q = M.T @ (K[:, None] * M) + 1e-3 * np.eye(M.shape[1])
    ns = len(in_state)
    for r in out_rows:
        R = np.linalg.solve(Q, (-1.0 if as_written else 1.0) * (M.T @ (K * Y[:, r])))
        A[r, in_state] += R[:ns]
        B[r, in_ctrl] += R[ns:-1]
        C[r] += R[-1]
    return A, B, C
