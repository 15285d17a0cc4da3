"""lmpc_solve_batch_warm: the reference's warm-start inputs (X_optm_ref, U_optm_ref; racing_mpc.cpp:293-305) used as an ACTIVE-SET
start (round 5, VERDICT r4 missing #3 / item 5).  What is checked: a warm solve returns the optimum the cold solve returns --
whatever the plan: the optimum itself, the shifted previous plan of a closed loop, or noise -- and takes one or two polish rounds
when the plan is good; kernel and twin run the same attempt; the closed loop ends where the cold loop ends."""
import numpy as np
import pytest
import torch

from oracle import cbind, params as P, scenario as S
from tolerances import TOL_DU, TOL_TWIN

pytestmark = pytest.mark.gpu
SX, SU = P.SCALE_X[:, None, None], P.SCALE_U[:, None, None]


def _np(d):
    return {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in d.items() if not k.startswith("_")}


def _err(a, b, ok):
    return max(np.abs((a["X_optm"] - b["X_optm"]) / SX)[..., ok].max(), np.abs((a["U_optm"] - b["U_optm"]) / SU)[..., ok].max(),
               np.abs((a["dU_optm"] - b["dU_optm"]) / SU)[..., ok].max())


def _periods(sv, tr, inp, n):
    """n closed-loop periods from `inp` (cold solves): the inputs of period n + 1, whose X_ref / U_ref are the shifted previous plan"""
    out = sv.alloc_outputs(inp["x_ic"].shape[1])
    nxt = inp
    for _ in range(n):
        sv.solve(nxt, out)
        good = (out["status"] == 0)[None, :]
        u0 = torch.where(good, out["U_optm"][:, 0, :], nxt["U_ref"][:, 0, :]).contiguous()
        xn = sv.plant_step(tr, nxt["x_ic"].clone(), u0, 0.0125, 2)
        nxt = sv.shift(tr, nxt, out, 0.025, speed_scale=0.9)
        nxt["x_ic"], nxt["u_ic"] = xn, u0
    return nxt


def _start(pkg, N, B, seed=11):
    tr = pkg.workloads.synthetic_track("barc")
    sv = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), device=0)
    rng = np.random.default_rng(seed)
    s0 = rng.uniform(0, tr["L"], B)
    x = np.stack([s0, rng.uniform(-0.1, 0.1, B), rng.normal(0, 0.03, B), 0.7 * S.track_lookup(tr["vel"], s0, tr["L"]), rng.normal(0, 0.02, B),
                  rng.normal(0, 0.1, B)], axis=1)
    inp = sv.prepare(tr, x.T.copy(), 0.025, speed_scale=0.9)
    inp["u_ic"] = torch.zeros((2, B), dtype=torch.float64, device="cuda")
    return tr, sv, inp


@pytest.mark.parametrize("N,B", [(20, 8192), (60, 4096)])
def test_default_rounds_go_by_the_batch_size(pkg, N, B):
    """lmpc_set_warm_rounds(0): two rounds while the batch is less than four times what the device holds at once, four beyond.
    (MI355X: 256 CUs x 8 problems at N = 20, x 4 at N = 60.)"""
    tr, sv, inp = _start(pkg, N, B)
    assert B >= 4 * 256 * sv.launch_info()["resident_problems_per_cu"]
    nxt = _periods(sv, tr, inp, 6)
    it = {}
    for r in (0, 2, 4):
        sv.set_warm_rounds(r)
        it[r] = sv.solve(nxt, warm=True)["iters"].cpu().numpy()
    assert np.array_equal(it[0], it[4]) and not np.array_equal(it[2], it[4])
    small = {k: (v[..., : B // 8].contiguous() if torch.is_tensor(v) and v.ndim and v.shape[-1] == B else v) for k, v in nxt.items()}
    for r in (0, 2):
        sv.set_warm_rounds(r)
        it[r] = sv.solve(small, warm=True)["iters"].cpu().numpy()
    assert np.array_equal(it[0], it[2])
    sv.close()


@pytest.mark.parametrize("N", [20, 40, 60])
def test_warm_from_the_optimum_from_noise_and_from_the_shifted_plan(pkg, N):
    B = 1024
    tr = pkg.workloads.synthetic_track("barc")
    cfg, veh = P.barc_tracking_mpc(N), P.barc_vehicle()
    sv = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), device=0)
    rng = np.random.default_rng(11)
    s0 = rng.uniform(0, tr["L"], B)
    x = np.stack([s0, rng.uniform(-0.1, 0.1, B), rng.normal(0, 0.03, B), 0.7 * S.track_lookup(tr["vel"], s0, tr["L"]), rng.normal(0, 0.02, B),
                  rng.normal(0, 0.1, B)], axis=1)
    inp = sv.prepare(tr, x.T.copy(), 0.025, speed_scale=0.9)
    inp["u_ic"] = torch.zeros((2, B), dtype=torch.float64, device="cuda")
    cold = _np(sv.solve(inp))
    ok = cold["status"] == 0
    assert ok.mean() > 0.99
    # (1) the plan IS the optimum: accepted at once, same answer
    w1 = _np(sv.solve(inp, warm={"X_optm_ref": torch.as_tensor(cold["X_optm"], device="cuda"), "U_optm_ref": torch.as_tensor(cold["U_optm"], device="cuda")}))
    assert (w1["status"][ok] == 0).all() and (w1["iters"][ok] <= 2).mean() > 0.98, np.bincount(w1["iters"][ok])
    assert _err(w1, cold, ok) < 1e-8
    # (2) noise: refused, the cold start takes over -- same statuses, same optimum
    noise = {"X_optm_ref": torch.randn((6, N, B), dtype=torch.float64, device="cuda"), "U_optm_ref": 0.01 * torch.randn((2, N - 1, B), dtype=torch.float64, device="cuda")}
    w2 = _np(sv.solve(inp, warm=noise))
    assert np.array_equal(w2["status"], cold["status"]) and _err(w2, cold, ok) < 1e-8
    assert (w2["iters"][ok] >= cold["iters"][ok]).all()          # (the rounds a refused attempt spent are counted)
    # (3) ten closed-loop periods later (the first plans after a cold start still change their active sets from period to
    # period: one period in, a quarter of the attempts is accepted; in the loop's steady state 95 %): the shifted plan, warm
    # against cold and against the twin's warm solve
    nxt = _periods(sv, tr, inp, 10)
    c3, w3 = _np(sv.solve(nxt)), _np(sv.solve(nxt, warm=True))
    ok3 = c3["status"] == 0
    assert np.array_equal(w3["status"] == 0, ok3)
    hit = ok3 & (w3["iters"] <= 2)
    print("N = %d: ten periods in, the warm attempt is accepted on %.3f of %d cars; iterations warm %.2f cold %.2f; warm vs cold %.1e"
          % (N, hit.sum() / ok3.sum(), ok3.sum(), w3["iters"][ok3].mean(), c3["iters"][ok3].mean(), _err(w3, c3, ok3)))
    assert hit.sum() / ok3.sum() > (0.8 if N <= 40 else 0.6) and _err(w3, c3, ok3) < TOL_TWIN
    tw = cbind.solve_batch(cfg, veh, _np(nxt), warm=True)
    assert np.array_equal(tw["status"] == 0, ok3)
    assert _err(w3, tw, ok3) < TOL_DU and (np.abs(w3["iters"][ok3] - tw["iters"][ok3]) == 0).mean() > 0.9
    # (4) lmpc_set_warm_rounds: more repair rounds accept more attempts, the answers stay the cold solve's, kernel and twin agree;
    # an accepted attempt reports at most the rounds allowed, a refused one more
    sv.set_warm_rounds(4)
    w5 = _np(sv.solve(nxt, warm=True))
    tw5 = cbind.solve_batch(cfg, veh, _np(nxt), warm=True, warm_rounds=4)
    sv.set_warm_rounds(0)
    assert np.array_equal(w5["status"] == 0, ok3) and _err(w5, c3, ok3) < TOL_TWIN
    hit5 = ok3 & (w5["iters"] <= 4)
    print("   four rounds allowed: accepted on %.3f (two rounds: %.3f)" % (hit5.sum() / ok3.sum(), hit.sum() / ok3.sum()))
    assert hit5.sum() >= hit.sum() and np.array_equal(w5["iters"][hit], w3["iters"][hit])
    assert _err(w5, tw5, ok3) < TOL_DU and (np.abs(w5["iters"][ok3] - tw5["iters"][ok3]) == 0).mean() > 0.9
    w0 = _np(sv.solve(nxt, warm=True))          # back on the default: the two-round result again
    assert np.array_equal(w0["iters"], w3["iters"])
    with pytest.raises(pkg.LmpcError, match="lmpc_set_warm_rounds"):
        sv.set_warm_rounds(5)
    sv.close()


def test_closed_loop_warm_ends_where_the_cold_loop_ends(pkg):
    B, steps = 512, 80
    tr = pkg.workloads.synthetic_track("barc")
    rng = np.random.default_rng(3)
    s0 = rng.uniform(0, tr["L"], B)
    x0 = torch.as_tensor(np.stack([s0, rng.uniform(-0.1, 0.1, B), rng.normal(0, 0.03, B), 0.7 * S.track_lookup(tr["vel"], s0, tr["L"]),
                                   rng.normal(0, 0.02, B), rng.normal(0, 0.1, B)]), dtype=torch.float64, device="cuda")
    u0 = torch.zeros((2, B), dtype=torch.float64, device="cuda")
    res = {}
    for warm in (False, True):
        sv = pkg.Solver(pkg.presets.barc_tracking_mpc(20), pkg.presets.barc_vehicle(), device=0)
        res[warm] = pkg.closed_loop.run(sv, tr, x0, u0, steps=steps, speed_scale=0.9, warm=warm)
        torch.cuda.synchronize()
        sv.close()
    d = (res[True]["x"] - res[False]["x"]).abs().cpu().numpy() / P.SCALE_X[:, None]
    same = (res[True]["n_fail"] == 0) & (res[False]["n_fail"] == 0)
    print("closed loop, %d cars x %d periods: warm hit rate %.3f; final states warm vs cold %.1e (cars without a failed solve: %d)"
          % (B, steps, res[True]["warm_hit_rate"], d[:, same.cpu().numpy()].max(), int(same.sum())))
    assert res[True]["warm_hit_rate"] > 0.85
    assert d[:, same.cpu().numpy()].max() < 1e-6
    assert torch.equal(res[True]["n_fail"] > 0, res[False]["n_fail"] > 0)


def test_warm_is_refused_for_a_learning_handle(pkg):
    sv = pkg.Solver(dict(pkg.presets.barc_lmpc(20, 3)), pkg.presets.barc_vehicle(), device=0)
    tr = pkg.workloads.synthetic_track("barc")
    x, u = pkg.workloads.sample_initial_states("barc", 8, tr["L"], [-0.01, -0.3], [0.01, 0.3], seed=0)
    inp = sv.prepare(tr, x.T.copy(), 0.025)
    inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device="cuda")
    with pytest.raises(pkg.LmpcError, match="tracking problem only"):
        sv.solve(inp, warm=True)
    sv.close()
