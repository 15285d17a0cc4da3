"""Every problem of a full batch at the horizons the reference SHIPS, kernel against serial twin.

The YAML files of the reference run the tracking controller at N = 40 / 60 / 80 (barc_tracking_mpc.param.yaml:7,
iac_car_tracking_mpc.param.yaml) and the learning controller at N = 40 / 60 (barc_lmpc.param.yaml, iac_car_lmpc.param.yaml);
until round 5 the GPU suite held those horizons to the twin on 24 - 48 problems only, and the builder's own full-size record
(profiles/r04_fullsize_parity.txt) showed what that hid: 1.1e-6 at N = 80 and 1.7e-5 at learning N = 60, problems whose polish was
refused one multiplier step short of convergence (fixed in round 5: csrc/lmpc_solve_kernel.hip `polish_limits<double>`).  Here:
batches of 4096 (IAC: the 8192 of configs[3]'s per-GPU share), the bench's own distributions, same assertions as
test_gpu_path.py::test_full_batch_properties -- statuses equal, X / U / dU of every problem within 1e-6 (scaled), iteration counts
equal on >= 90 %.  The twin is held to the dense optimum by tests/test_dense_fixtures.py (CPU) and scratch/r5/acc_dense.py."""
import numpy as np
import pytest
import torch

from oracle import cbind, params as P
from parity import assert_same_iterations
from tolerances import TOL_DU, TOL_TWIN

pytestmark = pytest.mark.gpu
SX, SU = P.SCALE_X[:, None, None], P.SCALE_U[:, None, None]


def _np(d):
    return {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in d.items()}


def _compare(tag, o, tw, min_solved):
    ok = (o["status"] == 0) & (tw["status"] == 0)
    assert (o["status"] == tw["status"]).all(), (tag, np.bincount(o["status"], minlength=3), np.bincount(tw["status"], minlength=3),
                                                 np.nonzero(o["status"] != tw["status"])[0][:8])
    assert ok.mean() >= min_solved, (tag, np.bincount(o["status"], minlength=3))
    ex = np.abs((o["X_optm"] - tw["X_optm"]) / SX).max(axis=(0, 1))[ok]
    eu = np.abs((o["U_optm"] - tw["U_optm"]) / SU).max(axis=(0, 1))[ok]
    ed = np.abs((o["dU_optm"] - tw["dU_optm"]) / SU).max(axis=(0, 1))[ok]
    di = np.abs(o["iters"][ok] - tw["iters"][ok])
    print("%s: %d problems, status %s; max scaled |dX| %.1e |dU| %.1e |d(dU)| %.1e; iterations equal %.4f, max difference %d, mean %.2f"
          % (tag, ok.size, np.bincount(o["status"], minlength=3).tolist(), ex.max(), eu.max(), ed.max(), (di == 0).mean(), di.max(),
             o["iters"][ok].mean()))
    worst = np.nonzero(ok)[0][np.argsort(np.maximum(np.maximum(ex, eu), ed))[-3:]]
    assert ex.max() < TOL_TWIN and eu.max() < TOL_TWIN and ed.max() < TOL_DU, (tag, ex.max(), eu.max(), ed.max(), worst)
    assert_same_iterations(o["iters"][ok], tw["iters"][ok])


@pytest.mark.parametrize("kind,N,B", [("barc", 40, 4096), ("barc", 60, 4096), ("barc", 80, 4096), ("iac", 40, 8192), ("iac", 80, 4096)])
def test_tracking_full_batch_against_twin(pkg, kind, N, B):
    iac = kind == "iac"
    tr = pkg.workloads.synthetic_track("putnam" if iac else "barc")
    if iac:
        x, u = pkg.workloads.sample_initial_states("putnam", B, tr["L"], [-10.0, -0.314159], [5.0, 0.314159], seed=1)
        pc, pv, oc, ov = pkg.presets.iac_tracking_mpc(N), pkg.presets.iac_vehicle(), P.iac_tracking_mpc(N), P.iac_vehicle()
    else:
        x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], seed=0)
        pc, pv, oc, ov = pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), P.barc_tracking_mpc(N), P.barc_vehicle()
    sv = pkg.Solver(pc, pv, device=0)
    inp = sv.prepare(tr, x.T.copy(), 0.025)
    inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device="cuda")
    o = _np(sv.solve(inp))
    tw = cbind.solve_batch(oc, ov, _np(inp))
    _compare("%s tracking N = %d" % (kind.upper(), N), o, tw, 0.999)
    sv.close()


@pytest.mark.parametrize("N,B", [(40, 4096), (60, 4096)])
def test_learning_full_batch_against_twin(pkg, N, B):
    tr = pkg.workloads.synthetic_track("barc")
    cfg = dict(pkg.presets.barc_lmpc(N, 5))
    laps = pkg.workloads.synthetic_laps(tr, 5)
    x, u = pkg.workloads.sample_states_near_laps(laps, B, tr["L"], seed=0)
    sv = pkg.Solver(cfg, pkg.presets.barc_vehicle(), device=0)
    sv.reserve(B)
    inp = sv.prepare(tr, x.T.copy(), 0.025)
    inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device="cuda")
    sv.set_safe_set(laps, tr["L"])
    s_last, s0, L = inp["X_ref"][0, -1], inp["x_ic"][0], tr["L"]
    kk = (s0 - s_last).abs() + L / 2
    q = torch.stack([s_last + (kk - torch.fmod(kk, L)) * torch.sign(s0 - s_last), inp["X_ref"][1, -1]]).contiguous()
    ss_x, ss_j, _ = sv.ss_query(q)
    out = sv.alloc_outputs(B)
    out["convex_combi_optm"] = torch.zeros((160, B), dtype=torch.float64, device="cuda")
    o = _np(sv.solve(inp, out, ss_x=ss_x, ss_j=ss_j))
    tw = cbind.solve_batch(P.barc_lmpc(N, 5), P.barc_vehicle(), _np(inp), ss_x.cpu().numpy(), ss_j.cpu().numpy())
    _compare("BARC learning N = %d, 160 points" % N, o, tw, 0.999)
    lam = o["convex_combi_optm"][:, o["status"] == 0]
    assert np.abs(lam.sum(0) - 1.0).max() < 1e-8 and lam.min() > -1e-9
    sv.close()
