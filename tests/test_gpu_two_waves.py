"""GPU: the two-wavefronts-per-problem kernels (csrc/lmpc_solve_w2.hip.h; round 6) -- the fp64 tracking problem
from N = 24 on with the inequality rows dealt to 128 lanes and the Riccati chains on one of the two waves.  A second kernel for the
same algorithm, so it is held to everything the one-wave kernel is held to:
  * the DENSE oracle's committed optima at the shipped horizons, every problem (tests/golden/dense_*.npz);
  * the serial twin on 1024 problems per horizon: answers, statuses, iteration counts;
  * the one-wave kernel on the same batch: same statuses, same iteration counts, answers to 1e-9;
  * bitwise reproducibility from launch to launch;
and the dispatch: the library takes these kernels from N = 41 on by itself, lmpc_set_waves_per_problem forces either."""
import numpy as np
import pytest
import torch

import dense_cases as DC
from oracle import cbind, params as P
from parity import assert_same_iterations, per_problem_err
from tolerances import TOL_DU, TOL_MEDIAN, TOL_TWIN, TOL_XU

pytestmark = pytest.mark.gpu
GOLD = DC.__file__.rsplit("/", 1)[0] + "/golden"


def _np(out):
    return {k: v.cpu().numpy() for k, v in out.items() if hasattr(v, "cpu")}


def _solver(pkg, family, N, waves):
    if family == "iac":
        sv = pkg.Solver(pkg.presets.iac_tracking_mpc(N), pkg.presets.iac_vehicle(), device=0)
    else:
        sv = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), device=0)
    sv.set_waves_per_problem(waves)
    return sv


@pytest.mark.parametrize("name", ["barc_tracking_n40", "barc_tracking_n60", "barc_tracking_n80", "iac_tracking_n40", "iac_tracking_n80"])
def test_two_wave_kernel_against_dense_fixture_every_problem(pkg, name):
    family, N = DC.CASES[name][0], DC.CASES[name][1]
    d = np.load(f"{GOLD}/dense_{name}.npz")
    fx = {k: d[k] for k in d.files}
    cfg, veh, inp, _, _ = DC.build(pkg, name)
    np.testing.assert_allclose(DC.digest(inp), fx["digest"], rtol=1e-11, atol=0)
    sv = _solver(pkg, family, N, 2)
    assert sv.launch_info("f64")["threads_per_problem"] == 128
    o = _np(sv.solve(inp))
    sv.close()
    assert (o["status"] == 0).all(), (name, np.nonzero(o["status"])[0])
    exu, ed = per_problem_err(o, fx)
    print("%s, two waves per problem: %d problems, kernel vs dense X/U max %.1e median %.1e, dU max %.1e; iterations mean %.2f"
          % (name, exu.size, exu.max(), np.median(exu), ed.max(), o["iters"].mean()))
    assert exu.max() < TOL_XU and ed.max() < TOL_DU and np.median(exu) < TOL_MEDIAN


@pytest.mark.parametrize("family,N", [("trk", 24), ("trk", 40), ("trk", 41), ("trk", 60), ("trk", 64), ("trk", 65), ("trk", 80), ("trk", 81), ("iac", 80)])
def test_two_wave_kernel_against_the_twin_and_the_one_wave_kernel(pkg, family, N):
    B = 1024
    tr = pkg.workloads.synthetic_track("putnam" if family == "iac" else "barc")
    if family == "iac":
        cfg, veh = P.iac_tracking_mpc(N), P.iac_vehicle()
        x, u = pkg.workloads.sample_initial_states("putnam", B, tr["L"], [-10.0, -0.314159], [5.0, 0.314159], seed=1)
    else:
        cfg, veh = P.barc_tracking_mpc(N), P.barc_vehicle()
        x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], seed=0)
    sv = _solver(pkg, family, N, 2)
    inp = sv.prepare(tr, x.T.copy(), 0.025)
    inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device="cuda")
    o2 = _np(sv.solve(inp))
    o2b = _np(sv.solve(inp))
    assert all(np.array_equal(o2[k], o2b[k]) for k in ("X_optm", "U_optm", "dU_optm", "status", "iters")), "not reproducible from launch to launch"
    sv.set_waves_per_problem(1)
    assert sv.launch_info("f64")["threads_per_problem"] == 64
    o1 = _np(sv.solve(inp))
    sv.close()
    # statuses: equal, except that a borderline polish acceptance may fall either way between two kernels whose sweeps differ in the
    # last bits (tests/dispatch_sweep.py's rule: at most two problems per case, never INFEASIBLE against anything else)
    differ = np.nonzero(o1["status"] != o2["status"])[0]
    assert differ.size <= 2 and not ((o1["status"][differ] == 2) | (o2["status"][differ] == 2)).any(), (differ, o1["status"][differ], o2["status"][differ])
    ok = (o1["status"] == 0) & (o2["status"] == 0)
    e12 = np.abs((o2["X_optm"] - o1["X_optm"]) / P.SCALE_X[:, None, None]).max(axis=(0, 1))[ok].max()
    tw = cbind.solve_batch(cfg, veh, {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in inp.items()})
    assert ((tw["status"] == 0) != (o2["status"] == 0)).sum() <= 2, (np.where((tw["status"] == 0) != (o2["status"] == 0))[0])
    ok = ok & (tw["status"] == 0)
    sel = {k: o2[k][..., ok] for k in ("X_optm", "U_optm", "dU_optm")}
    exu, ed = per_problem_err(sel, {k: tw[k][..., ok] for k in sel})
    print("%s N = %d, two waves: vs one wave %.1e (iterations equal on %.3f), vs twin X/U %.1e dU %.1e; solved %d of %d"
          % (family, N, e12, (o1["iters"] == o2["iters"]).mean(), exu.max(), ed.max(), ok.sum(), B))
    assert e12 < 1e-8 and exu.max() < TOL_TWIN and ed.max() < TOL_TWIN
    assert (o1["iters"] == o2["iters"]).mean() > 0.98
    assert_same_iterations(o2["iters"][ok], tw["iters"][ok])


def test_dispatch_takes_two_waves_from_n41_on(pkg):
    for N, threads in ((20, 64), (40, 64), (41, 128), (60, 128), (64, 128), (65, 128), (80, 128)):
        sv = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), device=0)
        assert sv.launch_info("f64")["threads_per_problem"] == threads, (N, sv.launch_info("f64"))
        sv.set_waves_per_problem(2)
        assert sv.launch_info("f64")["threads_per_problem"] == (128 if N >= 24 else 64)
        sv.close()
    lm = pkg.Solver(pkg.presets.barc_lmpc(80, 3), pkg.presets.barc_vehicle(), device=0)     # the learning problem: one wave
    lm.set_waves_per_problem(2)
    assert lm.launch_info("f64")["threads_per_problem"] == 64
    with pytest.raises(pkg.LmpcError, match="lmpc_set_waves_per_problem"):
        lm.set_waves_per_problem(3)
    lm.close()


def test_full_dynamics_through_the_two_wave_kernel(pkg):
    """lmpc_solve_full_dynamics_batch (the IPOPT role: sequential QPs over the same kernels) at the horizon barc_tracking_mpc.param.yaml
    ships: the library's choice (two waves per problem from N = 41 on) against the one-wave kernel forced -- the same first-order
    points from the node's first-solve state (racing_mpc_node.cpp:210-235,299-314)."""
    from oracle import scenario as S

    N, B = 60, 16
    veh, cfg = P.barc_vehicle(), P.barc_tracking_mpc(N)
    tr = pkg.workloads.synthetic_track("barc")
    x = np.zeros((B, 6))
    x[:, 0] = np.linspace(0.0, tr["L"], B, endpoint=False)
    x[:, 3] = 1.5
    inp = S.cold_start_inputs(cfg, veh, tr, x, np.zeros((B, 2)), 0.025)
    res = {}
    for waves in (0, 1):
        sv = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), device=0)
        sv.set_waves_per_problem(waves)
        assert sv.launch_info("f64")["threads_per_problem"] == (128 if waves == 0 else 64)
        res[waves] = _np(sv.solve_full_dynamics(inp, max_sqp=40, tol=1e-9))
        sv.close()
    a, b = res[0], res[1]
    conv = (a["status"] == 0) & (a["sqp_move"] <= 1e-9)
    assert conv.all(), (np.bincount(a["status"]), a["sqp_move"])
    assert np.array_equal(a["status"], b["status"]) and a["defect"].max() < 1e-7
    e = np.abs((a["X_optm"] - b["X_optm"]) / P.SCALE_X[:, None, None]).max()
    print("full dynamics, N = 60: two waves vs one wave X %.1e; QPs per problem %.1f / %.1f" % (e, a["sqp_iters"].mean(), b["sqp_iters"].mean()))
    assert e < 1e-6
