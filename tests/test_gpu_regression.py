"""Error-dynamics regression kernel (BASELINE config 5) against the CPU restatement.  Needs an MI355X."""
import numpy as np
import pytest
import torch

from oracle import params as P, qp as Q, regression as R, scenario as S
from test_regression_oracle import perturbed_plant_laps, planted_pairs, synthetic_lap

pytestmark = pytest.mark.gpu


def recorded_laps(veh, n_laps=3, n=150):
    """Pseudo-laps with the statistics the feature space needs (speeds, yaw rates, inputs around a driving point)."""
    return [synthetic_lap(veh, n, 20 + l) for l in range(n_laps)]


@pytest.mark.parametrize("as_written", [False, True])
def test_regression_matches_the_restatement_on_A_B_g(pkg, as_written):
    veh, cfg = P.barc_vehicle(), P.barc_tracking_mpc(10)
    solver = pkg.Solver(pkg.presets.barc_tracking_mpc(10), pkg.presets.barc_vehicle(), device=0)
    tr = pkg.workloads.synthetic_track("barc")
    u_lo, u_hi, _, _ = Q.effective_bounds(cfg, veh)
    B = 24
    x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], u_lo, u_hi, 31)
    inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
    laps = recorded_laps(veh)
    h = 0.6
    solver.set_regression_laps(laps, in_state=(3, 4, 5), in_ctrl=(0, 1), out_rows=(3, 4, 5), dist_max=h, as_written=as_written)
    A0, B0, g0 = solver.linearize(inp)
    A, Bm, g = solver.regress(inp, A0.clone(), B0.clone(), g0.clone())
    A0, B0, g0, A, Bm, g = (t.cpu().numpy() for t in (A0, B0, g0, A, Bm, g))
    n_touched = 0
    for b in range(0, B, 3):
        for i in range(cfg.N - 1):
            Ar, Br, gr = R.regress(veh, laps, (3, 4, 5), (0, 1), (3, 4, 5), h, inp["X_ref"][:, i, b], inp["U_ref"][:, i, b],
                                   A0[:, :, i, b], B0[:, :, i, b], g0[:, i, b], as_written=as_written)
            n_touched += int(np.abs(Ar - A0[:, :, i, b]).max() > 0)
            sc = 1.0 + np.abs(Ar).max()
            assert np.abs(A[:, :, i, b] - Ar).max() < 1e-9 * sc
            assert np.abs(Bm[:, :, i, b] - Br).max() < 1e-9 * (1.0 + np.abs(Br).max())
            assert np.abs(g[:, i, b] - gr).max() < 1e-9 * (1.0 + np.abs(gr).max())
    assert n_touched > 10  # the bandwidth actually catches lap samples on this workload


def test_regression_recovers_a_planted_error_model_and_enters_the_solve(pkg):
    veh = P.barc_vehicle()
    solver = pkg.Solver(pkg.presets.barc_tracking_mpc(10), pkg.presets.barc_vehicle(), device=0)
    tr = pkg.workloads.synthetic_track("barc")
    gain = np.array([[0.02, 0.0, 0.01, 0.5, 0.0, 0.001], [0.0, -0.03, 0.0, 0.0, 0.02, 0.0], [0.01, 0.0, 0.0, 0.0, 0.1, -0.002]])
    laps = planted_pairs(veh, 400, 5, gain)
    x = np.tile(np.array([1.0, 0.0, 0.0, 1.6, 0.0, 0.0]), (8, 1))
    inp = solver.prepare(tr, x.T.copy(), 0.025)
    inp["u_ic"] = torch.zeros((2, 8), dtype=torch.float64, device="cuda")
    base = solver.solve(inp)["X_optm"].cpu().numpy()
    solver.set_regression_laps(laps, dist_max=3.0)
    A0, B0, g0 = solver.linearize(inp)
    A, Bm, g = solver.regress(inp, A0.clone(), B0.clone(), g0.clone())
    dA = (A - A0).cpu().numpy()[3:, 3:6, 0, 0]
    assert np.allclose(dA, gain[:, :3], atol=2e-2)           # the planted error model, with its sign
    with_reg = solver.solve(inp)
    assert (with_reg["status"].cpu().numpy() == 0).all()
    assert np.abs(with_reg["X_optm"].cpu().numpy() - base).max() > 1e-6   # the corrected model reaches the QP
    solver.set_regression_laps([])
    assert np.abs(solver.solve(inp)["X_optm"].cpu().numpy() - base).max() == 0.0


def test_regression_reduces_the_one_step_error_of_a_perturbed_plant_on_the_device(pkg):
    """The default regression in front of the solve does what it is for: laps recorded on a plant with less
    grip and more mass, linearisation about recorded samples -- the corrected (A, B, g) predicts the plant's next state
    better than the nominal linearisation; the reference's literal signs (as_written) predict it worse."""
    veh = P.barc_vehicle()
    laps, plant = perturbed_plant_laps(veh)
    idx = list(range(0, 600, 25))
    B, N = len(idx), 3
    solver = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), device=0)
    X = np.zeros((6, N, B)); U = np.zeros((2, N - 1, B)); K = np.zeros((N, B))
    for b, j in enumerate(idx):
        X[:, :, b] = laps[j][0][0][:, None]
        U[:, :, b] = laps[j][1][0][:, None]
        K[:, b] = laps[j][2][0]
    inp = {"X_ref": X, "U_ref": U, "T_ref": np.full((N - 1, B), 0.03), "curvatures": K}
    truth = np.stack([laps[j][0][1] for j in idx], axis=1)        # [6][B]

    def one_step_error(A, Bm, g):
        A, Bm, g = (t.cpu().numpy() for t in (A, Bm, g))
        pred = np.einsum("rcb,cb->rb", A[:, :, 0], X[:, 0]) + np.einsum("rcb,cb->rb", Bm[:, :, 0], U[:, 0]) + g[:, 0]
        return np.median(np.abs(pred - truth)[3:].max(axis=0))

    A0, B0, g0 = solver.linearize(inp)
    e0 = one_step_error(A0, B0, g0)
    solver.set_regression_laps(laps, dist_max=0.6)
    e1 = one_step_error(*solver.regress(inp, A0.clone(), B0.clone(), g0.clone()))
    solver.set_regression_laps(laps, dist_max=0.6, as_written=True)
    ew = one_step_error(*solver.regress(inp, A0.clone(), B0.clone(), g0.clone()))
    assert e1 < 0.4 * e0 and ew > 5 * e0, (e0, e1, ew)
