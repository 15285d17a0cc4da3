"""The shipped horizons on the UNCLIPPED cold-start distribution, against the dense optimum.

barc_tracking_mpc.param.yaml ships n = 60, iac_car_tracking_mpc n = 80, barc_lmpc n = 40 (reference
src/launch/racing_lmpc_launch/param/racing_mpc/*.yaml); the QP is racing_mpc.cpp:106-201.  The sample contains starts
below 1 m/s, where the RK4 map of the tyre dynamics has |eig A| ~ 15-25 per step -- the problems a plain Riccati
recursion lost in round 1 (1.6 % at N = 40, 7 % at N = 60, 16 % at N = 80 reported infeasible).  Fixtures:
tests/golden/make_golden_long.py.  The serial twin is held to the same contract on the CPU, the kernel on the GPU.
"""
import numpy as np
import pytest

from oracle import cbind, params as P, qp as Q, scenario as S
from parity import assert_contract, assert_same_iterations

HORIZONS = [40, 60, 80]


def regenerate(pkg, N, B, seed):
    veh, cfg = P.barc_vehicle(), P.barc_tracking_mpc(N)
    tr = pkg.workloads.synthetic_track("barc")
    u_lo, u_hi, _, _ = Q.effective_bounds(cfg, veh)
    x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], u_lo, u_hi, seed)
    return veh, cfg, S.cold_start_inputs(cfg, veh, tr, x, u, 0.025), x


@pytest.mark.parametrize("N", HORIZONS)
def test_fixture_is_the_unclipped_sample(pkg, golden, N):
    """The committed inputs ARE the bench's cold-start distribution (nothing clipped away) and include the slow starts."""
    g = golden(f"qp_barc_tracking_long_n{N}")
    F = g["x_ic"].shape[1]
    veh, cfg, inp, x = regenerate(pkg, N, 256, 0)   # the fixture holds the first F problems of the 256-sample
    assert np.array_equal(g["x_ic"], x.T[:, :F])
    assert (x[:F, 3] < 1.0).sum() >= 5 and x[:F, 3].min() < 0.6
    assert np.allclose(g["X_ref"], inp["X_ref"][:, :, :F], rtol=0, atol=1e-9)
    assert (g["dense_status"] == 0).all()


@pytest.mark.parametrize("N", HORIZONS)
def test_twin_meets_the_contract_on_unclipped_long_horizons(golden, N):
    g = golden(f"qp_barc_tracking_long_n{N}")
    veh, cfg = P.barc_vehicle(), P.barc_tracking_mpc(N)
    out = cbind.solve_batch(cfg, veh, g)
    frac = assert_contract(out, g, g["margin"], g["certified"], who="twin")
    print(f"N = {N}: degenerate fraction {frac:.3f}")


@pytest.mark.parametrize("N", HORIZONS)
def test_twin_solves_every_problem_the_dense_solver_solves(pkg, golden, N):
    st = golden(f"long_status_n{N}")
    veh, cfg, inp, _ = regenerate(pkg, N, int(st["batch"]), int(st["seed"]))
    out = cbind.solve_batch(cfg, veh, inp)
    assert (out["status"][st["dense_status"] == 0] == 0).all(), np.where(out["status"] != 0)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("N", HORIZONS)
def test_kernel_meets_the_contract_on_unclipped_long_horizons(pkg, golden, N):
    """Status 0 on every problem and 1e-6 against the DENSE optimum on every problem (a quarter to a third of these are
    degenerate by the oracle's margin: the polish makes no difference between them and the strict ones)."""
    g = golden(f"qp_barc_tracking_long_n{N}")
    solver = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), device=0)
    out = {k: v.cpu().numpy() for k, v in solver.solve(g).items() if hasattr(v, "cpu")}
    frac = assert_contract(out, g, g["margin"], g["certified"])
    print(f"N = {N}: degenerate fraction {frac:.3f}")
    twin = cbind.solve_batch(P.barc_tracking_mpc(N), P.barc_vehicle(), g)
    assert_same_iterations(out["iters"], twin["iters"])


@pytest.mark.gpu
@pytest.mark.parametrize("N", HORIZONS)
def test_kernel_solves_every_problem_the_dense_solver_solves(pkg, golden, N):
    st = golden(f"long_status_n{N}")
    veh, cfg, inp, _ = regenerate(pkg, N, int(st["batch"]), int(st["seed"]))
    solver = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), device=0)
    status = solver.solve(inp)["status"].cpu().numpy()
    assert (status[st["dense_status"] == 0] == 0).all(), np.where(status != 0)[0]


@pytest.mark.gpu
def test_headline_batch_status_parity(pkg):
    """BASELINE configs[1] as bench.py runs it (batch 4096, seed 0, device-side cold start): whatever the kernel does
    not report optimal, the dense solver must not be able to solve either (and the other way round on a sample)."""
    B = 4096
    veh, cfg = P.barc_vehicle(), P.barc_tracking_mpc(20)
    solver = pkg.Solver(pkg.presets.barc_tracking_mpc(20), pkg.presets.barc_vehicle(), device=0)
    tr = pkg.workloads.synthetic_track("barc")
    u_lo, u_hi = [max(cfg.u_min[0], -0.015), max(cfg.u_min[1], -0.314159)], [min(cfg.u_max[0], 0.015), min(cfg.u_max[1], 0.314159)]
    x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], u_lo, u_hi, seed=0)
    inp = solver.prepare(tr, x.T.copy(), 0.025)
    inp["u_ic"] = u.T.copy()
    out = solver.solve(inp)
    status = out["status"].cpu().numpy()
    npinp = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in inp.items()}
    bad = np.where(status != 0)[0]
    assert len(bad) <= 8, len(bad)
    for b in bad:
        y, info = Q.solve_dense(Q.build_qp(cfg, veh, S.problem(npinp, int(b))))
        assert info["status"] != 0, (int(b), "the dense solver finds an optimum the kernel did not", x[b])
    for b in range(0, B, 128):  # and a sample of the solved ones is solvable for the dense solver too
        if status[b] == 0:
            assert Q.solve_dense(Q.build_qp(cfg, veh, S.problem(npinp, b)))[1]["status"] == 0, b


# ---- the IAC learning controller as shipped: iac_car_lmpc.param.yaml, n = 60 (tests/golden/make_golden_iac_lmpc.py) ----
def _iac_lmpc_check(out, g, who):
    """Every problem solved and within the contract of the dense optimum.  (With the simplex rows' weight floored inside
    the Newton matrix -- round 1 -- these problems stopped 1e-3 .. 1e-2 away with status 0; the two-level elimination of the
    terminal block reaches 1e-12 on them.)"""
    from parity import per_problem_err
    from tolerances import TOL_XU
    e, ed = per_problem_err(out, g)
    st = np.asarray(out["status"])
    assert (st == 0).all(), (who, st)
    assert e.max() < TOL_XU and ed.max() < TOL_XU, (who, e, ed)      # degenerate or not: all of them
    lam = np.asarray(out["convex_combi_optm"])
    assert np.abs(lam.sum(0) - 1.0).max() < 1e-9 and lam.min() > -1e-12
    assert np.asarray(out["iters"]).max() <= 20


def test_twin_on_the_iac_learning_controller_golden(golden):
    g = golden("qp_iac_lmpc_n60")
    out = cbind.solve_batch(P.iac_lmpc(60, 3), P.iac_vehicle(), g, ss_x=g["ss_x"], ss_j=g["ss_j"])
    _iac_lmpc_check(out, g, "twin")


@pytest.mark.gpu
def test_kernel_on_the_iac_learning_controller_golden(pkg, golden):
    """Safe-set query kernel -> LMPC QP kernel with the IAC vehicle and iac_car_lmpc.param.yaml's weights at its N = 60."""
    import torch
    g = golden("qp_iac_lmpc_n60")
    solver = pkg.Solver(pkg.presets.iac_lmpc(60, 3), pkg.presets.iac_vehicle(), device=0)
    solver.set_safe_set(list(g["laps"]), float(g["L"]))
    ss_x, ss_j, nf = solver.ss_query(g["query"])
    assert np.array_equal(ss_x.cpu().numpy(), g["ss_x"]) and np.array_equal(ss_j.cpu().numpy(), g["ss_j"])
    B = g["x_ic"].shape[1]
    out = solver.alloc_outputs(B)
    out["convex_combi_optm"] = torch.zeros((96, B), dtype=torch.float64, device="cuda")
    o = {k: v.cpu().numpy() for k, v in solver.solve(g, out, ss_x=ss_x, ss_j=ss_j).items() if hasattr(v, "cpu")}
    _iac_lmpc_check(o, g, "kernel")
