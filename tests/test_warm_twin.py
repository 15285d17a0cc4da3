"""CPU: the twin's warm solve (oracle/c/lmpc_oracle.c, the restatement of lmpc_solve_batch_warm) against the committed dense optima.
The plan handed in is what the reference hands its solver as the initial guess (X_optm_ref / U_optm_ref, racing_mpc.cpp:293-305).

  * the plan is the dense optimum: the active-set attempt is accepted within two rounds on nearly every problem and the answer is
    the dense optimum to the cold solve's tolerance, every problem;
  * the plan is noise: the attempt is refused, the cold start takes over -- statuses and answers of the cold twin, and the rounds
    the refused attempt spent are counted in `iters`."""
import numpy as np
import pytest

import dense_cases as DC
from oracle import cbind
from parity import per_problem_err
from tolerances import TOL_DU, TOL_XU

GOLD = DC.__file__.rsplit("/", 1)[0] + "/golden"


@pytest.mark.parametrize("name", ["barc_tracking_n20", "iac_tracking_n40", "barc_tracking_n60"])
def test_twin_warm_from_the_dense_optimum_and_from_noise(pkg, name):
    d = np.load(f"{GOLD}/dense_{name}.npz")
    fx = {k: d[k] for k in d.files}
    cfg, veh, inp, ss_x, ss_j = DC.build(pkg, name)
    assert ss_x is None
    B = min(DC.CASES[name][2], 128)
    sl = {k: (np.ascontiguousarray(v[..., :B]) if isinstance(v, np.ndarray) and v.ndim and v.shape[-1] == DC.CASES[name][2] else v) for k, v in inp.items()}
    fxs = {k: fx[k][..., :B] for k in ("X_optm", "U_optm", "dU_optm")}
    cold = cbind.solve_batch(cfg, veh, sl)
    assert (cold["status"] == 0).all()

    plan = dict(sl, X_ref=np.ascontiguousarray(fxs["X_optm"]), U_ref=np.ascontiguousarray(fxs["U_optm"]))
    # (the warm entry takes the plan where the cold one takes the linearisation trajectory; the problem itself -- references,
    #  bounds, x_ic, u_ic -- is unchanged: cbind hands X_ref / U_ref of `warm_plan` as the plan)
    w = cbind.solve_batch(cfg, veh, sl, warm=True, warm_plan=plan)
    assert (w["status"] == 0).all()
    exu, ed = per_problem_err(w, fxs)
    print("%s: twin warm from the dense optimum: accepted within two rounds %.3f; vs dense X/U %.1e dU %.1e; iterations warm %.2f cold %.2f"
          % (name, (w["iters"] <= 2).mean(), exu.max(), ed.max(), w["iters"].mean(), cold["iters"].mean()))
    assert exu.max() < TOL_XU and ed.max() < TOL_DU
    assert (w["iters"] <= 2).mean() > 0.9

    # more rounds allowed (lmpc_set_warm_rounds): whatever was accepted within two rounds is unchanged, the rest may be accepted later
    w6 = cbind.solve_batch(cfg, veh, sl, warm=True, warm_plan=plan, warm_rounds=6)
    two = w["iters"] <= 2
    assert np.array_equal(w6["iters"][two], w["iters"][two]) and (w6["iters"] <= 6).sum() >= two.sum() and (w6["status"] == 0).all()
    exu, ed = per_problem_err(w6, fxs)
    assert exu.max() < TOL_XU and ed.max() < TOL_DU

    rng = np.random.default_rng(5)
    noise = dict(sl, X_ref=rng.normal(size=fxs["X_optm"].shape), U_ref=0.01 * rng.normal(size=fxs["U_optm"].shape))
    n = cbind.solve_batch(cfg, veh, sl, warm=True, warm_plan=noise)
    assert np.array_equal(n["status"], cold["status"])
    exu, ed = per_problem_err(n, cold)
    assert exu.max() < 1e-8 and ed.max() < 1e-8
    assert (n["iters"] >= cold["iters"]).all()


@pytest.mark.parametrize("case", ["barc_lmpc_n20_s160", "barc_lmpc_spec_n20_s160", "barc_lmpc_n20_s96"])
def test_twin_learning_warm_start_from_the_optimum(pkg, case):
    """The learning problem's warm start on the serial twin (lmpc_oracle_solve_range_warm_lam; the kernel's lmpc_solve_batch_warm_ss
    follows it line for line): the optimum as the plan, its simplex weights as convex_combi_optm_ref (racing_mpc.cpp:281) ->
    accepted by the active-set attempt within two rounds on >= 90 %, the cold solve's answer to 1e-8; a plan with noise on it is
    refused and costs the cold solve plus its rounds; without weights the call is the cold solve."""
    import dense_cases as DC
    from oracle import cbind, params as OP

    cfg, veh, inp, ss_x, ss_j = DC.build(pkg, case)
    n = 96
    inp = {k: (v[..., :n] if hasattr(v, "shape") and np.ndim(v) >= 1 and np.shape(v)[-1] >= n else v) for k, v in inp.items()}
    ss_x, ss_j = ss_x[..., :n], ss_j[..., :n]
    cold = cbind.solve_batch(cfg, veh, inp, ss_x, ss_j)
    ok = cold["status"] == 0
    assert ok.all()
    plan = {"X_ref": cold["X_optm"], "U_ref": cold["U_optm"], "lam": cold["convex_combi_optm"]}
    warm = cbind.solve_batch(cfg, veh, inp, ss_x, ss_j, warm=True, warm_plan=plan)
    e = np.abs((warm["X_optm"] - cold["X_optm"]) / OP.SCALE_X[:, None, None]).max(axis=(0, 1))
    print("%s: accepted %.3f, iterations warm %.2f cold %.2f, worst %.1e" % (case, (warm["iters"] <= 4).mean(), warm["iters"].mean(), cold["iters"].mean(), e.max()))
    assert (warm["status"] == 0).all() and (warm["iters"] <= 4).mean() >= 0.9 and e.max() < 1e-8
    assert np.abs(warm["convex_combi_optm"] - cold["convex_combi_optm"]).max() < 1e-6
    rng = np.random.default_rng(0)
    noisy = {"X_ref": cold["X_optm"] + rng.normal(0, 1e-3, cold["X_optm"].shape) * OP.SCALE_X[:, None, None],
             "U_ref": cold["U_optm"] + rng.normal(0, 1e-3, cold["U_optm"].shape) * OP.SCALE_U[:, None, None], "lam": cold["convex_combi_optm"]}
    w2 = cbind.solve_batch(cfg, veh, inp, ss_x, ss_j, warm=True, warm_plan=noisy)
    assert np.array_equal(w2["status"], cold["status"]) and np.abs(w2["X_optm"] - cold["X_optm"]).max() < 1e-9 and (w2["iters"] >= cold["iters"]).all()
    w3 = cbind.solve_batch(cfg, veh, inp, ss_x, ss_j, warm=True, warm_plan={"X_ref": cold["X_optm"], "U_ref": cold["U_optm"]})
    assert np.array_equal(w3["X_optm"], cold["X_optm"]) and np.array_equal(w3["iters"], cold["iters"])
