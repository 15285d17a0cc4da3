"""Oracle self-checks for the dynamics (CPU)."""
from pathlib import Path

import numpy as np
import pytest

from oracle import cbind, dynamics as D, params as P

GOLD = Path(__file__).parent / "golden"


def _random_points(veh, rng, n, fast):
    x = np.stack([rng.uniform(0, 15, n), rng.uniform(-0.3, 0.3, n), rng.normal(0, 0.1, n),
                  rng.uniform(15, 70, n) if fast else rng.uniform(0.5, 3.0, n), rng.normal(0, 0.05, n),
                  rng.normal(0, 0.2, n)], -1)
    u = np.stack([rng.normal(0, 1.0 if fast else 0.005, n), rng.normal(0, 0.1, n)], -1)
    k = rng.uniform(-0.03, 0.05, n) if fast else rng.uniform(-0.3, 0.9, n)
    return x, u, k


def test_analytic_jacobian_matches_complex_step():
    rng = np.random.default_rng(0)
    for veh, fast in ((P.barc_vehicle(), False), (P.iac_vehicle(), True)):
        x, u, k = _random_points(veh, rng, 300, fast)
        A, B, g = D.rk4_jacobian_cs(x, u, k, 0.025, veh)
        A2, B2, g2, _ = D.rk4_jacobian_analytic(x, u, k, 0.025, veh)
        assert np.abs(A - A2).max() <= 1e-11 * max(1.0, np.abs(A).max())
        assert np.abs(B - B2).max() <= 1e-11 * max(1.0, np.abs(B).max())
        assert np.abs(g - g2).max() <= 1e-10 * max(1.0, np.abs(g).max())


@pytest.mark.parametrize("integrator", ["rk4", "euler"])
def test_c_oracle_linearisation_matches_complex_step(integrator):
    """Both integrators of the model (single_track_planar_model.cpp:357-368): RK4, and Euler x + dt f (utils.cpp:110-123)."""
    import dataclasses
    rng = np.random.default_rng(1)
    veh, cfg = dataclasses.replace(P.barc_vehicle(), integrator=integrator), P.barc_tracking_mpc(8)
    B = 40
    x, u, k = _random_points(veh, rng, B * 8, False)
    inp = {"X_ref": x.reshape(8, B, 6).transpose(2, 0, 1).copy(), "U_ref": u.reshape(8, B, 2)[:7].transpose(2, 0, 1).copy(),
           "T_ref": np.full((7, B), 0.025), "curvatures": k.reshape(8, B).copy()}
    A, Bm, g = cbind.linearize_batch(cfg, veh, inp)
    Ar, Br, gr = D.rk4_jacobian_cs(inp["X_ref"][:, :7].transpose(1, 2, 0), inp["U_ref"].transpose(1, 2, 0),
                                   inp["curvatures"][:7], 0.025, veh)
    assert np.abs(A.transpose(2, 3, 0, 1) - Ar).max() <= 1e-11 * np.abs(Ar).max()
    assert np.abs(Bm.transpose(2, 3, 0, 1) - Br).max() <= 1e-11 * np.abs(Br).max()
    assert np.abs(g.transpose(1, 2, 0) - gr).max() <= 1e-10 * max(1.0, np.abs(gr).max())


def test_known_input_of_reference_model_test():
    # test_single_track_planar_model.cpp:68-79 evaluates x=[0,0,0,40,1,0.1], u=[0,0.1], k=0.1 and
    # prints x_dot without asserting it.  Frenet rows have closed forms independent of the tyres.
    veh = P.barc_vehicle()
    f = D.f_continuous(np.array([0.0, 0, 0, 40, 1, 0.1]), np.array([0.0, 0.1]), 0.1, veh)
    assert abs(f[0] - 40.0) < 1e-12 and abs(f[1] - 1.0) < 1e-12 and abs(f[2] - (0.1 - 0.1 * 40.0)) < 1e-12
    # u_lon = 0: fd = fb = 0, so only rolling resistance and tyre forces act
    assert np.all(np.isfinite(f))


def test_reference_recorded_laps_are_consistent_with_the_dynamics():
    """Plausibility pin: replay the reference's recorded BARC laps (its simulator stepped the same
    model at 10 ms with asynchronous control updates, so agreement is to ~1e-3, not rounding)."""
    veh = P.barc_vehicle()
    base = GOLD / "barc_ss" / "ss_lap_1"
    x, u, k, t = (np.loadtxt(f"{base}_{s}.txt") for s in "xukt")
    errs = []
    for i in range(5, 300):
        dt = t[i + 1] - t[i]
        nsub = max(int(round(dt / 0.01)), 1)
        xx = x[i].copy()
        for j in range(nsub):
            xx = D.rk4(xx, u[i + 1], k[i] + (k[i + 1] - k[i]) * j / nsub, dt / nsub, veh)
        errs.append(np.abs(xx - x[i + 1]))
    mean = np.mean(errs, axis=0)
    assert mean[0] < 3e-3 and mean[1] < 5e-4 and mean[2] < 2e-3 and mean[3] < 5e-3 and mean[4] < 5e-3


def test_align_abscissa():
    L = 15.0
    assert abs(D.align_abscissa(0.5, 14.8, L) - 15.5) < 1e-12
    assert abs(D.align_abscissa(14.9, 0.2, L) - (-0.1)) < 1e-12
    assert abs(D.align_abscissa(3.0, 4.0, L) - 3.0) < 1e-12


def test_euler_is_one_slope():
    import dataclasses
    veh = dataclasses.replace(P.barc_vehicle(), integrator="euler")
    x, u, k = np.array([1.0, 0.05, 0.02, 2.0, 0.03, 0.4]), np.array([0.004, 0.1]), 0.2
    assert np.allclose(D.rk4(x, u, k, 0.025, veh), x + 0.025 * D.f_continuous(x, u, k, veh), rtol=0, atol=1e-15)
    A, B, g = D.rk4_jacobian_cs(x, u, k, 0.025, veh)
    Fx = (A - np.eye(6)) / 0.025
    assert np.abs(Fx[:, 0]).max() == 0.0 and abs(Fx[1, 2] - (x[3] * np.cos(x[2]) - x[4] * np.sin(x[2]))) < 1e-12
