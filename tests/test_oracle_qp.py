"""Oracle checks for the QP: golden vectors carry KKT certificates; the C twin reproduces them (CPU)."""
import numpy as np
import pytest

from conftest import scaled_err
from oracle import cbind, params as P, qp as Q, scenario as S
from parity import assert_contract
from tolerances import TOL_MEDIAN

CASES = [("qp_barc_tracking_n20", P.barc_vehicle, lambda: P.barc_tracking_mpc(20)),
         ("qp_barc_tracking_n10", P.barc_vehicle, lambda: P.barc_tracking_mpc(10)),
         ("qp_iac_tracking_n40", P.iac_vehicle, lambda: P.iac_tracking_mpc(40))]


@pytest.mark.parametrize("name,veh,cfg", CASES)
def test_golden_solutions_are_certified_optima(golden, name, veh, cfg):
    g = golden(name)
    veh, cfg = veh(), cfg()
    assert g["certified"].all()
    stat, eq, ineq, comp = g["kkt_cert"]
    assert stat.max() < 1e-9 and eq.max() < 1e-9 and ineq.max() < 1e-10 and comp.max() < 1e-9
    # re-certify two problems from scratch: solver-independent NNLS multipliers on the rebuilt QP
    for b in (0, g["x_ic"].shape[1] - 1):
        qp = Q.build_qp(cfg, veh, S.problem(g, b))
        y = Q.pack(qp, g["X_optm"][:, :, b], g["U_optm"][:, :, b], g["dU_optm"][:, :, b], sigma=g["sigma"][b])
        c = Q.kkt_certificate(qp, y)
        assert c["stat"] < 1e-8 and c["eq"] < 1e-9 and c["ineq"] < 1e-9 and c["comp"] < 1e-8, c
        assert abs(qp.objective(y) - g["objective"][b]) < 1e-9 * (1 + abs(g["objective"][b]))


@pytest.mark.parametrize("name,veh,cfg", CASES)
def test_c_twin_matches_golden(golden, name, veh, cfg):
    g = golden(name)
    veh, cfg = veh(), cfg()
    out = cbind.solve_batch(cfg, veh, g)
    assert out["iters"].max() <= 30
    assert_contract(out, g, g["margin"], g["certified"], who="twin")
    per = np.abs((out["X_optm"] - g["X_optm"]) / P.SCALE_X[:, None, None]).max(axis=(0, 1))
    assert np.median(per) < TOL_MEDIAN
    # the objective is reproduced far more tightly than the (flat-direction) variables
    for b in range(0, g["x_ic"].shape[1], 5):
        qp = Q.build_qp(cfg, veh, S.problem(g, b))
        y = Q.pack(qp, out["X_optm"][:, :, b], out["U_optm"][:, :, b], out["dU_optm"][:, :, b], sigma=max(out["kkt"][3, b], 0.0))
        assert qp.objective(y) - g["objective"][b] < 1e-7 * (1 + abs(g["objective"][b]))
        assert np.abs(qp.A @ y - qp.b).max() < 1e-9
        assert (qp.C @ y - qp.d).max() < 1e-8


def test_cold_start_inputs_layout(pkg):
    veh, cfg = P.barc_vehicle(), P.barc_tracking_mpc(12)
    tr = pkg.workloads.synthetic_track("barc")
    u_lo, u_hi, _, _ = Q.effective_bounds(cfg, veh)
    x, u = pkg.workloads.sample_initial_states("barc", 5, tr["L"], u_lo, u_hi, 0)
    inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
    assert inp["X_ref"].shape == (6, 12, 5) and inp["U_ref"].shape == (2, 11, 5)
    assert np.allclose(inp["X_ref"][:, 0, :], x.T)
    assert np.all(np.abs(inp["vel_ref"] - inp["X_ref"][3]) <= cfg.max_vel_ref_diff + 1e-12)
    assert np.all(inp["bound_left"] > 0) and np.all(inp["bound_right"] < 0)


def test_infeasible_initial_state_is_reported():
    veh, cfg = P.barc_vehicle(), P.barc_tracking_mpc(10)
    from __graft_entry__ import load_package
    wl = load_package().workloads
    tr = wl.synthetic_track("barc")
    x = np.array([[1.0, 0.0, 0.0, 0.05, 0.0, 0.0], [1.0, 0.0, 0.0, 1.0, 0.0, 0.0]])  # vx < x_min[3] = 0.1
    inp = S.cold_start_inputs(cfg, veh, tr, x, np.zeros((2, 2)), 0.025)
    out = cbind.solve_batch(cfg, veh, inp)
    assert out["status"][0] == 2 and out["status"][1] == 0


def test_lmpc_golden_and_c_twin(golden):
    """LMPC terminal block (racing_mpc.cpp:479-522): certified dense optimum vs the structured C twin."""
    g = golden("qp_barc_lmpc_n20")
    veh, cfg = P.barc_vehicle(), P.barc_lmpc(20, 3)
    assert g["certified"].all() and g["kkt_cert"][0].max() < 1e-9
    out = cbind.solve_batch(cfg, veh, g, ss_x=g["ss_x"], ss_j=g["ss_j"])
    assert (out["status"] == 0).all() and out["iters"].max() <= 25
    assert scaled_err(out["X_optm"], g["X_optm"], P.SCALE_X) < 1e-6
    assert scaled_err(out["U_optm"], g["U_optm"], P.SCALE_U) < 1e-6
    assert scaled_err(out["dU_optm"], g["dU_optm"], P.SCALE_U) < 1e-5
    lam = out["convex_combi_optm"]
    assert np.abs(lam.sum(0) - 1.0).max() < 1e-9 and lam.min() > -1e-12


def test_zero_components_of_the_hull_slack_weight(golden):
    """convex_hull_slack with zero entries (racing_mpc.cpp:497-499): those components of the hull slack are free -- the
    terminal state is tied to the convex hull only in the weighted components.  Twin against the dense optimum."""
    import dataclasses
    g = golden("qp_barc_lmpc_n20")
    veh = P.barc_vehicle()
    cfg = dataclasses.replace(P.barc_lmpc(20, 3), convex_hull_slack=np.array([40.0, 0.0, 4.0, 40.0, 0.0, 4.0]))
    out = cbind.solve_batch(cfg, veh, g, ss_x=g["ss_x"], ss_j=g["ss_j"], b1=6)
    assert (out["status"][:6] == 0).all()
    for b in range(6):
        qp = Q.build_qp(cfg, veh, S.problem(g, b), ss_x=g["ss_x"][:, :, b], ss_j=g["ss_j"][:, b])
        y, info = Q.solve_dense(qp)
        assert info["status"] == 0
        ex = qp.split(y)
        assert np.abs((out["X_optm"][:, :, b] - ex["X_optm"]) / P.SCALE_X[:, None]).max() < 2e-6, b
        assert np.abs((out["U_optm"][:, :, b] - ex["U_optm"]) / P.SCALE_U[:, None]).max() < 2e-6, b
        assert np.abs(ex["X_optm"][:, :, ] - g["X_optm"][:, :, b]).max() > 1e-6      # and it is a different problem


def test_hard_convex_hull_equality_twin_against_dense():
    """All-zero convex_hull_slack (racing_mpc.cpp:500-502: x_T = SS lambda, no slack).  Dense QP: the residual pinned to
    zero by six equality rows.  Twin: the penalty limit LMPC_HARD_HULL_WEIGHT of its terminal elimination -- within 1e-7 of
    the equality-constrained optimum wherever that exists; not OPTIMAL where the hull cannot be reached."""
    import dataclasses
    import lmpc_scenario as LS
    veh, cfg, tr, laps, inp, q = LS.make(24, 5)
    ss_x, ss_j, _ = LS.oracle_safe_set(cfg, laps, q)
    hard = dataclasses.replace(cfg, convex_hull_slack=np.zeros(6))
    tw = cbind.solve_batch(hard, veh, inp, ss_x, ss_j)
    n_ok = 0
    for b in range(24):
        qp = Q.build_qp(hard, veh, S.problem(inp, b), ss_x=ss_x[:, :, b], ss_j=ss_j[:, b])
        try:
            y, info = Q.solve_dense(qp)
        except np.linalg.LinAlgError:
            info = {"status": 9}
        if info["status"] != 0:
            assert tw["status"][b] != 0, b
            continue
        ex = qp.split(y)
        assert tw["status"][b] == 0, b
        assert np.abs((tw["X_optm"][:, :, b] - ex["X_optm"]) / P.SCALE_X[:, None]).max() < 2e-7, b
        assert np.abs((tw["U_optm"][:, :, b] - ex["U_optm"]) / P.SCALE_U[:, None]).max() < 2e-7, b
        eps = tw["X_optm"][:, -1, b] - ss_x[:, :, b] @ tw["convex_combi_optm"][:, b]
        assert np.abs(eps / P.SCALE_X).max() < 1e-7
        n_ok += 1
    assert n_ok >= 15
