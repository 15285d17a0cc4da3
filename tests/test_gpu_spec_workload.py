"""GPU: the LEARNING workload bench.py quotes since round 5 (`--workload lmpc --lmpc-data spec` = SURVEY.md 8(d) config 3 as written:
five laps recorded by the tracking loop at speed scales 0.80 .. 1.0, configs[1]'s random x0) under test at the sizes the bench runs it
(VERDICT r5 item 1a).  Until round 6 every learning fixture and full-size test drew states NEAR the laps (`sample_states_near_laps`),
a friendlier distribution than the one the BENCH_r05 line was measured on.

  * the laps: a fresh recording on this GPU is the committed data file (tests/golden/spec_laps.npz) the dense fixture was solved on;
  * configs[2] (4096, fp64): every problem the kernel does not report optimal is one the DENSE solver cannot solve either (and a
    sample of the solved ones is solvable for it), the first 512 are in tests/golden/dense_barc_lmpc_spec_n20_s160.npz
    (tests/test_gpu_dense_fixtures.py compares them one by one);
  * configs[4]'s share of one GPU (32768, mixed precision, regression on and off): status parity with fp64, the stated accuracy
    (tests/tolerances.py: TOL_F32 at the 99.99 % quantile, TOL_F32_WORST for every problem), and the fp64 failures against the dense
    solver on the same (regressed) stage models."""
import numpy as np
import pytest
import torch

import dense_cases as DC
from oracle import params as P, qp as Q, scenario as S
from parity import per_problem_err
from tolerances import TOL_F32, TOL_F32_WORST

pytestmark = pytest.mark.gpu
KEYS = ("X_optm", "U_optm", "dU_optm")
DEV = "cuda"


def _np(out):
    return {k: v.cpu().numpy() for k, v in out.items() if hasattr(v, "cpu")}


def _setup(pkg, B, regression=False):
    tr = pkg.workloads.synthetic_track("barc")
    laps = DC.spec_laps()
    cfgd = pkg.presets.barc_lmpc(20, 5)
    sv = pkg.Solver(cfgd, pkg.presets.barc_vehicle(), device=0)
    sv.set_safe_set(laps, tr["L"])
    if regression:   # as bench.py --regression: sample pairs of a plant with 15 % less grip around the stored laps
        pv = dict(pkg.presets.barc_vehicle())
        pv["mu"] *= 0.85
        plant = pkg.Solver(cfgd, pv, device=0)
        reg_laps = pkg.workloads.regression_sample_pairs(
            tr, laps, lambda xa, ua: plant.plant_step(tr, torch.as_tensor(xa.T.copy(), device=DEV), torch.as_tensor(ua.T.copy(), device=DEV),
                                                      0.03).cpu().numpy().T)
        plant.close()
        sv.set_regression_laps(reg_laps, dist_max=0.6)
    x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], seed=0)   # bench.py, rank 0
    inp = sv.prepare(tr, x.T.copy(), 0.025)
    inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=DEV)
    q = torch.as_tensor(DC.ss_query_point({k: inp[k].cpu().numpy() for k in ("X_ref", "x_ic")}, tr["L"]), device=DEV).contiguous()
    ss_x, ss_j, _ = sv.ss_query(q)
    return sv, tr, inp, ss_x, ss_j


def _solve(sv, inp, ss_x, ss_j, mixed):
    B = inp["x_ic"].shape[-1]
    out = sv.alloc_outputs(B)
    out["convex_combi_optm"] = torch.zeros((160, B), dtype=torch.float64, device=DEV)
    return _np(sv.solve(inp, out, ss_x=ss_x, ss_j=ss_j, mixed=mixed))


def _dense_status(cfg, veh, npinp, ss_x, ss_j, b, lin=None):
    qp = Q.build_qp(cfg, veh, S.problem(npinp, int(b)), ss_x=ss_x[:, :, b], ss_j=ss_j[:, b], lin=lin)
    try:
        return Q.solve_dense(qp)[1]["status"]
    except np.linalg.LinAlgError:
        return -1


def test_recorded_laps_are_the_committed_data(pkg):
    """tests/golden/spec_laps.npz is what closed_loop.record_laps produces on this GPU (so the dense fixture built on the file pins
    the workload bench.py runs): same lap lengths, samples within 1e-9 (the kernels are bitwise reproducible on one box; across
    boxes the FMA contraction of a rebuilt library may move a last bit through a 200-period closed loop)."""
    tr = pkg.workloads.synthetic_track("barc")
    sv = pkg.Solver(pkg.presets.barc_tracking_mpc(20), pkg.presets.barc_vehicle(), device=0)
    laps = pkg.closed_loop.record_laps(sv, tr)
    sv.close()
    ref = DC.spec_laps()
    assert [lap.shape for lap in laps] == [lap.shape for lap in ref], ([lap.shape for lap in laps], [lap.shape for lap in ref])
    worst = max(float(np.abs(a - b).max()) for a, b in zip(laps, ref))
    print("spec laps: %s samples, fresh recording against the committed file max |d| %.1e" % ([lap.shape[0] for lap in laps], worst))
    assert worst < 1e-9, worst


def test_configs2_as_benched_status_parity_with_the_dense_solver(pkg):
    """BENCH_r05: solved_fraction 0.99976 on configs[2] -- 1 of 4096.  Whatever the kernel does not report optimal the dense solver must
    not be able to solve either; and the GPU's safe-set query + cold start are the oracle's on the fixture's 512 problems."""
    B = 4096
    sv, tr, inp, ss_x, ss_j = _setup(pkg, B)
    o = _solve(sv, inp, ss_x, ss_j, False)
    sv.close()
    cfg, veh = P.barc_lmpc(20, 5), P.barc_vehicle()
    npinp = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in inp.items()}
    sx, sj = ss_x.cpu().numpy(), ss_j.cpu().numpy()
    _, _, inp_o, ssx_o, ssj_o = DC.build(pkg, "barc_lmpc_spec_n20_s160")     # the CPU route the fixture was built by
    assert np.array_equal(sx[:, :, :512], ssx_o) and np.array_equal(sj[:, :512], ssj_o)
    # (same problems, not the same bits: the device's cold-start rollout and the oracle's restatement of it agree to rounding per
    #  step, and below 1 m/s the RK4 step map amplifies that ~25x per knot -- 2.5e-6 of the digest on 5 % of the problems; the
    #  fixture test feeds the kernel the CPU-built inputs, so parity there is on identical bits)
    np.testing.assert_allclose(DC.digest({k: npinp[k][..., :512] for k in DC.INPUT_KEYS}), DC.digest(inp_o), rtol=1e-4, atol=0)
    bad = np.where(o["status"] != 0)[0]
    print("configs[2] as benched: %d of %d not optimal: %s status %s iters %s; mean iterations %.2f"
          % (bad.size, B, bad.tolist(), o["status"][bad].tolist(), o["iters"][bad].tolist(), o["iters"].mean()))
    assert bad.size <= 8, bad.size
    for b in bad:
        assert _dense_status(cfg, veh, npinp, sx, sj, b) != 0, (int(b), "the dense solver finds an optimum the kernel did not")
    for b in range(0, B, 256):
        if o["status"][b] == 0:
            assert _dense_status(cfg, veh, npinp, sx, sj, b) == 0, b


@pytest.mark.parametrize("regression", [True, False])
def test_configs4_share_as_benched_every_problem(pkg, regression):
    """configs[4]'s share of one GPU on the workload bench.py quotes it on: 32768 problems, mixed precision against fp64 on the same
    (regressed) stage models.  Status parity both ways; accuracy as tests/tolerances.py states it for this workload -- TOL_F32 at the
    99.99 % quantile, TOL_F32_WORST for every problem: three of the 32768 pass the fp32 KKT test 1.2 .. 3.4e-3 from the fp64 answer
    (ill-conditioned blends of safe-set points; profiles/r06_mixed_tail_spec.txt); and every fp64 failure (BENCH_r05: 10 of 32768) is a
    problem the dense solver cannot solve either."""
    B = 32768
    sv, tr, inp, ss_x, ss_j = _setup(pkg, B, regression)
    o64, om = _solve(sv, inp, ss_x, ss_j, False), _solve(sv, inp, ss_x, ss_j, True)
    assert (o64["status"] == 0).mean() > 0.999, np.bincount(o64["status"])
    lost = np.where((o64["status"] == 0) & (om["status"] != 0))[0]
    assert lost.size == 0, ("solved in fp64, not by the mixed entry", lost[:8], om["status"][lost[:8]])
    assert not (om["status"] == 3).any()
    both = (o64["status"] == 0) & (om["status"] == 0)
    e, ed = per_problem_err({k: om[k][..., both] for k in KEYS}, {k: o64[k][..., both] for k in KEYS})
    qs = np.quantile(e, [0.5, 0.99, 0.9999])
    print("configs[4] share as benched (%s): %d problems, mixed vs fp64 X/U median %.1e 99%% %.1e 99.99%% %.1e max %.1e (%d above 1e-3) | dU max %.1e | "
          "status fp64 %s mixed %s" % ("regression" if regression else "no regression", B, qs[0], qs[1], qs[2], e.max(), (e > TOL_F32).sum(), ed.max(),
                                      np.bincount(o64["status"], minlength=4).tolist(), np.bincount(om["status"], minlength=4).tolist()))
    assert qs[2] < TOL_F32 and qs[1] < 1e-4, qs
    assert e.max() < TOL_F32_WORST and (e > TOL_F32).sum() <= 8, (e.max(), (e > TOL_F32).sum())
    assert ed.max() < TOL_F32_WORST / 0.025
    lam = om["convex_combi_optm"][:, om["status"] == 0]
    assert np.abs(lam.sum(0) - 1.0).max() < 1e-8 and lam.min() > -1e-9
    # the fp64 failures against the dense solver, on the stage models the kernel solved on (the regression's correction included)
    bad = np.where(o64["status"] != 0)[0]
    assert bad.size <= 24, bad.size
    cfg, veh = P.barc_lmpc(20, 5), P.barc_vehicle()
    npinp = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in inp.items()}
    sx, sj = ss_x.cpu().numpy(), ss_j.cpu().numpy()
    A, Bm, g = sv.linearize(inp)
    if regression:
        sv.regress(inp, A, Bm, g)
    A, Bm, g = A.cpu().numpy(), Bm.cpu().numpy(), g.cpu().numpy()
    for b in bad:
        lin = (np.ascontiguousarray(A[..., b].transpose(2, 0, 1)), np.ascontiguousarray(Bm[..., b].transpose(2, 0, 1)), np.ascontiguousarray(g[..., b].T))
        assert _dense_status(cfg, veh, npinp, sx, sj, b, lin=lin) != 0, (int(b), "the dense solver finds an optimum the kernel did not")
    sv.close()
