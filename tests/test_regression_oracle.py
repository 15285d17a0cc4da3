"""Error-dynamics regression (BASELINE config 5, safe_set.cpp:56-114,182-245): properties of the CPU restatement.
The reference holds no test or golden vector for this query (parity unpinned, see oracle/regression.py)."""
import numpy as np

from oracle import params as P
from oracle import regression as R
from oracle.dynamics import rk4


def synthetic_lap(veh, n=60, seed=0, gain=None):
    """A smooth pseudo-lap (states drawn around a slow driving point, not a rollout)."""
    rng = np.random.default_rng(seed)
    x = np.array([0.0, 0.0, 0.0, 1.6, 0.0, 0.0]) + rng.normal(0, 1, (n, 6)) * np.array([0.5, 0.05, 0.05, 0.3, 0.05, 0.3])
    u = np.stack([rng.uniform(-0.005, 0.005, n), rng.uniform(-0.1, 0.1, n)], axis=1)
    return x, u, rng.uniform(-0.3, 0.3, n), np.arange(n) * 0.03


def planted_pairs(veh, n, seed, gain, as_written=False):
    """n two-sample laps (x_a, x_b) with x_b = nominal step + gain [vx, vy, w, u0, u1, 1] on the rows (vx, vy, w):
    the regression sees exactly that error.  The nominal step is the one the chosen variant evaluates: dt = +30 ms, or
    (as_written) dt = t_a - t_b < 0 as process_lap_data hands it to the model."""
    xs, us, ks, _ = synthetic_lap(veh, n, seed)
    laps = []
    for j in range(n):
        z = np.concatenate([xs[j, 3:6], us[j], [1.0]])
        xb = rk4(xs[j], us[j], float(ks[j]), -0.03 if as_written else 0.03, veh) + np.concatenate([np.zeros(3), gain @ z])
        laps.append((np.stack([xs[j], xb]), np.stack([us[j], us[j]]), np.array([ks[j], ks[j]]), np.array([0.0, 0.03])))
    return laps


def test_residuals_are_the_nominal_model_error():
    veh = P.barc_vehicle()
    x, u, k, t = synthetic_lap(veh, 30, 1)
    y = R.lap_residuals(veh, x, u, k, t)
    assert y.shape == (29, 6)
    yw = R.lap_residuals(veh, x, u, k, t, as_written=True)
    for j in (0, 7, 28):
        assert np.allclose(y[j], x[j + 1] - rk4(x[j], u[j], float(k[j]), float(t[j + 1] - t[j]), veh), atol=1e-15)
        assert np.allclose(yw[j], x[j + 1] - rk4(x[j], u[j], float(k[j]), float(t[j] - t[j + 1]), veh), atol=1e-15)


def test_no_candidate_leaves_the_model_untouched_and_far_points_do_not_count():
    veh = P.barc_vehicle()
    lap = synthetic_lap(veh, 40, 2)
    A, B, C = np.eye(6), np.ones((6, 2)), np.zeros(6)
    far = np.array([0.0, 0.0, 0.0, 50.0, 0.0, 0.0])
    A2, B2, C2 = R.regress(veh, [lap], (3, 4, 5), (0, 1), (3, 4, 5), 0.5, far, np.zeros(2), A, B, C)
    assert (A2 == A).all() and (B2 == B).all() and (C2 == C).all()


import pytest


@pytest.mark.parametrize("as_written", [False, True])
def test_weighted_ridge_normal_equations_and_both_signs(as_written):
    """R solves (M'KM + 1e-3 I) R = +M'K y (default) or -M'K y (the reference's literal sign) for every regressed row;
    rows not regressed and columns not in the feature lists are untouched."""
    veh = P.barc_vehicle()
    sgn = -1.0 if as_written else 1.0
    gain = np.array([[0.02, 0.0, 0.01, 0.5, 0.0, 0.001], [0.0, -0.03, 0.0, 0.0, 0.02, 0.0], [0.01, 0.0, 0.0, 0.0, 0.1, -0.002]])
    laps = planted_pairs(veh, 200, 3, gain, as_written)
    qx, qu = laps[20][0][0].copy(), laps[20][1][0].copy()
    A0, B0, C0 = np.zeros((6, 6)), np.zeros((6, 2)), np.zeros(6)
    h = 3.0
    A, B, C = R.regress(veh, laps, (3, 4, 5), (0, 1), (3, 4, 5), h, qx, qu, A0, B0, C0, as_written=as_written)
    assert np.abs(A[:3]).max() == 0 and np.abs(A[:, :3]).max() == 0 and np.abs(B[:3]).max() == 0 and np.abs(C[:3]).max() == 0
    Z = np.concatenate([np.concatenate([l[0][:-1, 3:6], l[1][:-1]], axis=1) for l in laps])
    Y = np.concatenate([R.lap_residuals(veh, *l, as_written=as_written) for l in laps])
    d = np.linalg.norm(Z - np.concatenate([qx[3:6], qu]), axis=1)
    m = d < h
    K = 0.75 / h * (1 - (d[m] / h) ** 2) ** 2
    M = np.concatenate([Z[m], np.ones((m.sum(), 1))], axis=1)
    for r in (3, 4, 5):
        Rr = np.concatenate([A[r, 3:6], B[r], [C[r]]])
        assert np.allclose((M.T * K) @ M @ Rr + 1e-3 * Rr, sgn * (M.T * K) @ Y[m, r], rtol=1e-9, atol=1e-12)
    # with the ridge small against M'KM the fit is the planted error model (minus it with the reference's sign)
    assert np.allclose(A[3:, 3:6], sgn * gain[:, :3], atol=2e-2) and np.allclose(B[3:, 1], sgn * gain[:, 4], atol=2e-2)


def perturbed_plant_laps(veh, n=600, seed=50):
    """Samples recorded on a plant that differs from the nominal model (less grip, more mass): n two-sample laps
    (x_a, x_b) with x_b the PLANT's 30 ms step from x_a, states drawn around a 2.5 m/s driving point (above the speeds
    where the 30 ms RK4 step of the tyre model is unstable)."""
    import dataclasses
    plant = dataclasses.replace(veh, mu=0.8 * veh.mu, m=1.1 * veh.m)
    xs, us, ks, _ = synthetic_lap(veh, n, seed)
    xs[:, 3] += 0.9
    laps = [(np.stack([xs[j], rk4(xs[j], us[j], float(ks[j]), 0.03, plant)]), np.stack([us[j], us[j]]),
             np.array([ks[j], ks[j]]), np.array([0.0, 0.03])) for j in range(n)]
    return laps, plant


def test_regression_reduces_the_one_step_error_of_a_perturbed_plant():
    """What the error-dynamics regression is for: recorded on a plant with 20 % less grip and 10 % more mass, the
    corrected linear model A x + B u + g predicts the plant's next state better than the nominal linearisation at the
    same point -- with the default signs.  As written upstream it predicts it worse."""
    from oracle.dynamics import rk4_jacobian_cs
    veh = P.barc_vehicle()
    laps, plant = perturbed_plant_laps(veh)
    err0, err1, errw = [], [], []
    for j in range(0, 600, 25):   # linearisation points = recorded samples; the fit is leave-nothing-out but local (h = 0.6)
        x, u, k = laps[j][0][0], laps[j][1][0], laps[j][2][:1]
        A, B, g = (a[0] for a in rk4_jacobian_cs(x[None], u[None], k, np.array([0.03]), veh))
        truth = laps[j][0][1]
        A1, B1, g1 = R.regress(veh, laps, (3, 4, 5), (0, 1), (3, 4, 5), 0.6, x, u, A, B, g)
        Aw, Bw, gw = R.regress(veh, laps, (3, 4, 5), (0, 1), (3, 4, 5), 0.6, x, u, A, B, g, as_written=True)
        err0.append(np.abs((A @ x + B @ u + g - truth)[3:]).max())
        err1.append(np.abs((A1 @ x + B1 @ u + g1 - truth)[3:]).max())
        errw.append(np.abs((Aw @ x + Bw @ u + gw - truth)[3:]).max())
    assert np.median(err1) < 0.4 * np.median(err0), (np.median(err0), np.median(err1))
    assert np.median(errw) > 5 * np.median(err0), (np.median(err0), np.median(errw))
