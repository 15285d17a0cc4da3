"""BASELINE configs[4]: the LEARNING problem with the mixed fp32/fp64 KKT (lmpc_solve_batch_mixed with learning = 1), and
the learning solve with the error-dynamics regression switched on.  Needs an MI355X.

What is mixed: the stage records, the Riccati factor and sweeps, and the stage rows run in fp32; the safe-set block --
simplex rows, the two-level terminal elimination (racing_mpc.cpp:484-504) -- the linearisation, the regression and every
array in HBM stay fp64.  Stated tolerance for an fp32 iteration: 1e-3 scaled (the order of the reference's
OSQP eps)."""
import numpy as np
import pytest
import torch

from oracle import params as P, qp as Q, scenario as S
from parity import per_problem_err
import lmpc_scenario as LS

pytestmark = pytest.mark.gpu
TOL_MIXED = 1e-3


def _solve(sv, inp, ss_x, ss_j, mixed):
    B = (inp["x_ic"].shape[1])
    out = sv.alloc_outputs(B)
    out["convex_combi_optm"] = torch.zeros((int(sv.config["num_ss_pts"]), B), dtype=torch.float64, device="cuda")
    return {k: v.cpu().numpy() for k, v in sv.solve(inp, out, ss_x=ss_x, ss_j=ss_j, mixed=mixed).items() if hasattr(v, "cpu")}


def test_mixed_learning_solve_matches_the_golden_vectors(pkg, golden):
    g = golden("qp_barc_lmpc_n20")
    sv = pkg.Solver(pkg.presets.barc_lmpc(20, 3), pkg.presets.barc_vehicle(), device=0)
    sv.set_safe_set(LS.load_laps(), LS.L_BARC_SS)
    ss_x, ss_j, _ = sv.ss_query(g["query"])
    o = _solve(sv, g, ss_x, ss_j, True)
    assert (o["status"] == 0).all(), o["status"]
    e, ed = per_problem_err(o, g)
    assert e.max() < TOL_MIXED and np.median(e) < 2e-4, e
    lam = o["convex_combi_optm"]
    assert np.abs(lam.sum(0) - 1.0).max() < 1e-9 and lam.min() > -1e-12     # the simplex rows are fp64
    assert o["iters"].mean() < 12


def _s160(pkg, B, seed=0):
    tr = pkg.workloads.synthetic_track("barc")
    laps = pkg.workloads.synthetic_laps(tr, 5)
    sv = pkg.Solver(pkg.presets.barc_lmpc(20, 5), pkg.presets.barc_vehicle(), device=0)
    sv.set_safe_set(laps, tr["L"])
    x, u = pkg.workloads.sample_states_near_laps(laps, B, tr["L"], seed=seed)
    inp = sv.prepare(tr, x.T.copy(), 0.025)
    inp["u_ic"] = torch.as_tensor(u.T.copy(), device="cuda")
    s_last, s0, L = inp["X_ref"][0, -1], inp["x_ic"][0], tr["L"]
    kk = (s0 - s_last).abs() + L / 2
    q = torch.stack([s_last + (kk - torch.fmod(kk, L)) * torch.sign(s0 - s_last), inp["X_ref"][1, -1]]).contiguous()
    ss_x, ss_j, _ = sv.ss_query(q)
    return sv, tr, laps, inp, ss_x, ss_j


def _dense_errors(pkg, n):
    sv, tr, laps, inp, ss_x, ss_j = _s160(pkg, 64)
    o = _solve(sv, inp, ss_x, ss_j, True)
    npinp = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in inp.items()}
    sx, sj = ss_x.cpu().numpy(), ss_j.cpu().numpy()
    cfg, veh = P.barc_lmpc(20, 5), P.barc_vehicle()
    errs = []
    for b in range(n):
        qp = Q.build_qp(cfg, veh, S.problem(npinp, b), ss_x=sx[:, :, b], ss_j=sj[:, b])
        y, info = Q.solve_dense(qp)
        assert info["status"] == 0 and o["status"][b] == 0, (b, info["status"], o["status"][b])
        ex = qp.split(y)
        errs.append(max(np.abs((o["X_optm"][:, :, b] - ex["X_optm"]) / P.SCALE_X[:, None]).max(),
                        np.abs((o["U_optm"][:, :, b] - ex["U_optm"]) / P.SCALE_U[:, None]).max()))
    return np.array(errs)


def test_mixed_learning_solve_against_dense_optima_with_160_points(pkg):
    """5 stored laps, 160 safe-set points: against the DENSE optimum."""
    e = _dense_errors(pkg, 32)
    print("mixed LMPC, S = 160, vs dense: median %.1e, 90 %% %.1e, max %.1e" % (np.median(e), np.percentile(e, 90), e.max()))
    assert np.median(e) < 1e-5 and e.max() < TOL_MIXED, np.sort(e)[-5:]


def test_mixed_learning_solve_every_problem_within_1e3(pkg):
    """Every problem of a full-size batch within the stated 1e-3 of the fp64 kernel's answer.  (Round 2 committed this as a
    strict xfail: median 1.4e-5, 99th percentile 3.3e-3, worst 1.0e-2 -- an fp32 interior point stops at mu ~ 2e-6, O(sqrt(mu))
    from the optimum where the active set is still ambiguous.  The fp32 polish verifies its answers; what it cannot verify,
    about a percent of a batch, is solved by the fp64 second pass.)"""
    sv, tr, laps, inp, ss_x, ss_j = _s160(pkg, 4096)
    o64, o32 = _solve(sv, inp, ss_x, ss_j, False), _solve(sv, inp, ss_x, ss_j, True)
    both = (o64["status"] == 0) & (o32["status"] == 0)
    assert both.mean() > 0.999
    e, _ = per_problem_err({k: o32[k][..., both] for k in ("X_optm", "U_optm", "dU_optm")},
                           {k: o64[k][..., both] for k in ("X_optm", "U_optm", "dU_optm")})
    print("mixed vs fp64, 4096 learning problems: median %.1e, 99 %% %.1e, max %.1e" % (np.median(e), np.percentile(e, 99), e.max()))
    assert e.max() < TOL_MIXED and np.percentile(e, 99) < 1e-4, (np.median(e), np.percentile(e, 99), e.max())


def test_mixed_one_pass_marks_what_it_cannot_verify(pkg):
    """lmpc_config.polish = 1: no fp64 second pass -- the problems whose fp32 polish was refused keep LMPC_SOLVE_UNVERIFIED (3);
    they are a small fraction, the accepted ones are within the stated tolerance, and the default two-pass solve reports
    none."""
    sv, tr, laps, inp, ss_x, ss_j = _s160(pkg, 2048)
    o64, two = _solve(sv, inp, ss_x, ss_j, False), _solve(sv, inp, ss_x, ss_j, True)
    cfg1 = dict(pkg.presets.barc_lmpc(20, 5)); cfg1["polish"] = 1
    sv1 = pkg.Solver(cfg1, pkg.presets.barc_vehicle(), device=0)
    sv1.set_safe_set(laps, tr["L"])
    one = _solve(sv1, inp, ss_x, ss_j, True)
    assert not (two["status"] == 3).any()
    marked = one["status"] == 3
    assert 0 < marked.mean() < 0.03, marked.mean()
    acc = (one["status"] == 0) & (o64["status"] == 0)
    e, _ = per_problem_err({k: one[k][..., acc] for k in ("X_optm", "U_optm", "dU_optm")},
                           {k: o64[k][..., acc] for k in ("X_optm", "U_optm", "dU_optm")})
    assert e.max() < TOL_MIXED, e.max()
    # the second pass writes the fp64 answers over the marked problems: bit-identical to the fp64 entry
    for k in ("X_optm", "U_optm", "dU_optm"):
        assert np.array_equal(two[k][..., marked], o64[k][..., marked]), k


@pytest.mark.parametrize("mixed", [False, True])
def test_learning_solve_with_the_regression_switched_on(pkg, mixed):
    """racing_mpc.cpp:479-522 + safe_set.cpp:182-245 together (config 5): laps recorded on a plant with less grip feed
    lmpc_set_regression_laps, and the learning solve runs with the corrected (A, B, g)."""
    import dataclasses
    from oracle.dynamics import rk4
    sv, tr, laps, inp, ss_x, ss_j = _s160(pkg, 512)
    base = _solve(sv, inp, ss_x, ss_j, mixed)
    veh = P.barc_vehicle()
    plant = dataclasses.replace(veh, mu=0.85 * veh.mu)
    reg_laps = []
    for lap in laps[-2:]:     # samples around the stored laps, successors from the PLANT (30 ms)
        n = lap.shape[0]
        rng = np.random.default_rng(n)
        x = lap + rng.normal(0, 1, lap.shape) * np.array([0.0, 0.02, 0.02, 0.1, 0.03, 0.2])
        u = np.stack([rng.uniform(-0.005, 0.005, n), rng.uniform(-0.15, 0.15, n)], axis=1)
        k = np.interp(x[:, 0], np.arange(tr["M"]) * tr["L"] / tr["M"], tr["curvature"], period=tr["L"])
        for j in range(0, n - 1, 2):
            x[j + 1] = rk4(x[j], u[j], float(k[j]), 0.03, plant)
        reg_laps.append((x, u, k, np.arange(n) * 0.03))
    sv.set_regression_laps(reg_laps, dist_max=0.6)
    with_reg = _solve(sv, inp, ss_x, ss_j, mixed)
    sv.set_regression_laps([])
    assert (with_reg["status"] == 0).mean() > 0.98, np.bincount(with_reg["status"])
    ok = (with_reg["status"] == 0) & (base["status"] == 0)
    assert np.abs(with_reg["X_optm"] - base["X_optm"])[:, :, ok].max() > 1e-4     # the corrected model reaches the QP
    again = _solve(sv, inp, ss_x, ss_j, mixed)
    assert np.array_equal(again["X_optm"], base["X_optm"])                        # and switching it off restores it


def test_learning_problem_at_full_size_in_fp64(pkg):
    """BASELINE configs[2] as bench.py runs it: batch 4096, 5 stored laps (160 safe-set points), N = 20, fp64.
    Size-independent properties on the whole batch, the contract against the DENSE optimum on a sample."""
    from parity import assert_contract, dense_reference
    B = 4096
    sv, tr, laps, inp, ss_x, ss_j = _s160(pkg, B)
    o = _solve(sv, inp, ss_x, ss_j, False)
    ok = o["status"] == 0
    assert ok.mean() > 0.999, np.bincount(o["status"])
    lam = o["convex_combi_optm"]
    assert lam[:, ok].min() > -1e-12 and np.abs(lam[:, ok].sum(0) - 1.0).max() < 1e-9          # the simplex (racing_mpc.cpp:489-491)
    A, Bm, g = (t.cpu().numpy() for t in sv.linearize(inp))
    X, U = o["X_optm"], o["U_optm"]
    pred = np.einsum("rcib,cib->rib", A, X[:, :-1]) + np.einsum("rcib,cib->rib", Bm, U) + g
    assert np.abs(pred - X[:, 1:])[:, :, ok].max() < 1e-8                                       # x_{i+1} = A x_i + B u_i + g
    npinp = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in inp.items()}
    sample = list(range(0, B, B // 12))[:12]
    ref, margin, certified, _, _ = dense_reference(P.barc_lmpc(20, 5), P.barc_vehicle(), npinp, sample, ss_x.cpu().numpy(), ss_j.cpu().numpy())
    frac = assert_contract({k: o[k][..., sample] for k in ("X_optm", "U_optm", "dU_optm", "status")}, ref, margin, certified,
                           who="learning kernel, 160 points")
    for b in np.where(~ok)[0][:4]:   # status parity: what the kernel gives up on, the dense solver does not solve either
        qp = Q.build_qp(P.barc_lmpc(20, 5), P.barc_vehicle(), S.problem(npinp, int(b)), ss_x=ss_x.cpu().numpy()[:, :, b], ss_j=ss_j.cpu().numpy()[:, b])
        _, info = Q.solve_dense(qp)
        assert info["status"] != 0, (int(b), int(o["status"][b]), "kernel failed where the dense solver succeeds")
    # EVERY problem of the batch against the serial twin: same statuses, answers within the twin tolerance
    from oracle import cbind
    from tolerances import TOL_TWIN
    tw = cbind.solve_batch(P.barc_lmpc(20, 5), P.barc_vehicle(), npinp, ss_x.cpu().numpy(), ss_j.cpu().numpy())
    both = ok & (tw["status"] == 0)
    assert (ok == (tw["status"] == 0)).mean() > 0.999, (np.bincount(tw["status"]), np.bincount(o["status"]))
    e, ed = per_problem_err({k: o[k][..., both] for k in ("X_optm", "U_optm", "dU_optm")}, {k: tw[k][..., both] for k in ("X_optm", "U_optm", "dU_optm")})
    print("learning problem at full size: solved %.5f, degenerate among the sample %.2f; against the twin on all %d: %.1e / dU %.1e"
          % (ok.mean(), frac, both.sum(), e.max(), ed.max()))
    assert e.max() < TOL_TWIN and ed.max() < TOL_TWIN, (e.max(), ed.max())


def test_second_pass_is_the_direct_fp64_kernel_bit_for_bit():
    """The fp64 second pass of the mixed solve (a persistent grid walking the list of marked problems and CALLING the
    solve) against the direct fp64 kernel, on whole batches (the debug build of the library -- the same sources with
    -DLMPC_DEBUG_HOOKS -- reads LMPC_DEBUG_CLEANUP_ALL=1 and hands every problem to the second pass): same status, iteration count and bits for every problem, for every (KQ, KS) the second pass is built for.
    Guards a failure seen in round 3: with the solve inlined under the persistent loop one instantiation computed garbage
    while the direct kernel was right."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    here = Path(__file__).resolve().parent
    dbg = here.parent / "racing-lmpc-ros2_amd" / "lib" / "liblmpc_hip_dbg.so"   # the hook exists in the debug build only (make debug)
    assert dbg.exists(), "liblmpc_hip_dbg.so not built: __graft_entry__.build()"
    env = dict(os.environ, LMPC_DEBUG_CLEANUP_ALL="1", LMPC_HIP_LIBRARY=str(dbg))
    r = subprocess.run([sys.executable, str(here / "second_pass_check.py")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [json.loads(line) for line in r.stdout.splitlines() if line.startswith("{")]
    assert len(rows) == 6, r.stdout
    for row in rows:
        assert row["same_bits"] and row["solved"] == row["solved_second_pass"] and row["solved"] >= 1020, row
