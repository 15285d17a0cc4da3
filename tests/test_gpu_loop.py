"""lmpc_loop_advance_batch: everything between two solves of a closed loop in one launch (the steps either
side of the solve -- racing_mpc_node.cpp:245-254, 322-332, 210-235; racing_simulator.cpp:46-69, 97-112).  It must be the composition
of the entry points it replaces: input selection, lmpc_plant_step_batch, lmpc_shift_batch / lmpc_prepare_failed_batch, and the
harness's statistics -- compared here bit for bit (one call; except the last knot's one-step rollout, to 2 ulp) and over whole
closed loops (to rounding)."""
import numpy as np
import pytest
import torch

from oracle import scenario as S

pytestmark = pytest.mark.gpu
KEYS = ("X_ref", "U_ref", "T_ref", "bound_left", "bound_right", "curvatures", "vel_ref")


def _states(tr, B, seed=3, fast=0.7):
    rng = np.random.default_rng(seed)
    s0 = rng.uniform(0, tr["L"], B)
    return np.stack([s0, rng.uniform(-0.1, 0.1, B), rng.normal(0, 0.03, B), fast * S.track_lookup(tr["vel"], s0, tr["L"]), rng.normal(0, 0.02, B),
                     rng.normal(0, 0.1, B)])


@pytest.mark.parametrize("restart", [True, False])
def test_one_call_is_the_composition_of_the_entry_points_it_replaces(pkg, restart):
    B, N, dt = 1024, 20, 0.025
    tr = pkg.workloads.synthetic_track("barc")
    sv = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), device=0)
    trk = sv.device_track(tr)
    x = torch.as_tensor(_states(tr, B, fast=1.3), dtype=torch.float64, device="cuda")      # (fast into the corners: some solves fail)
    u_prev = torch.zeros((2, B), dtype=torch.float64, device="cuda")
    inp = sv.prepare(trk, x, dt, speed_scale=0.9)
    inp["x_ic"], inp["u_ic"] = x, u_prev
    out = sv.solve(inp)
    out["status"][::7] = 1          # and some are declared failed, so that both branches are well populated
    ok = out["status"] == 0
    assert 0 < int((~ok).sum()) < B
    # the composition
    u_apply = torch.where(ok[None, :], out["U_optm"][:, 0, :], inp["U_ref"][:, 0, :]).contiguous()
    x_ref = sv.plant_step(trk, x.clone(), u_apply, dt / 2, 2)
    ds = x_ref[0] - x[0]
    dist_ref = torch.where(ds < -tr["L"] / 2, ds + tr["L"], ds)
    hb = float(sv.vehicle["b"]) / 2
    exc_ref = torch.maximum(torch.zeros(B, dtype=torch.float64, device="cuda"), torch.maximum(x_ref[1] + hb - inp["bound_left"][0], inp["bound_right"][0] - (x_ref[1] - hb)))
    nxt = sv.shift(trk, inp, out, dt, speed_scale=0.9)
    if restart:
        sv.prepare_failed(trk, x_ref, out["status"], nxt, dt, speed_scale=0.9)
    # the one call, on copies
    inp2 = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in inp.items()}
    x2, u2 = x.clone(), u_prev.clone()
    dist = torch.zeros(B, dtype=torch.float64, device="cuda")
    exc = torch.zeros(B, dtype=torch.float64, device="cuda")
    nf = torch.zeros(B, dtype=torch.int64, device="cuda")
    acc = torch.zeros((), dtype=torch.int64, device="cuda")
    sv.loop_advance(trk, inp2, out, x2, u2, dt, dt / 2, 2, speed_scale=0.9, restart_failed=restart, distance=dist, worst_excess=exc, n_fail=nf, n_accepted=acc)
    torch.cuda.synchronize()
    assert torch.equal(x2, x_ref) and torch.equal(u2, u_apply)
    assert torch.equal(dist, dist_ref) and torch.equal(exc, exc_ref) and torch.equal(nf, (~ok).to(torch.int64))
    assert int(acc) == int((ok & (out["iters"] <= 4)).sum())
    # Bit for bit, with one exception: the last knot of a shifted solution is one model step from the knot before it, and the
    # compiler contracts the inlined model differently in the two kernels (1 - 2 ulp on 476 of 6144 numbers, and through them on a
    # handful of the references sampled at that knot's abscissa).  Everything in front of the last knot, and every knot of the failed
    # cars (whole rollouts), is identical.
    for k in KEYS:
        a, r = inp2[k], nxt[k]
        if k in ("U_ref", "T_ref"):
            assert torch.equal(a, r), k
            continue
        assert torch.equal(a[..., :-1, :], r[..., :-1, :]), k
        assert torch.equal(a[..., -1, :][..., ~ok], r[..., -1, :][..., ~ok]), k
        assert torch.allclose(a[..., -1, :], r[..., -1, :], rtol=1e-14, atol=1e-15), (k, float((a[..., -1, :] - r[..., -1, :]).abs().max()))
    # the accumulators accumulate; NULL accumulators are allowed
    sv.loop_advance(trk, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in inp.items()}, out, x.clone(), u_prev.clone(), dt, dt / 2, 2, speed_scale=0.9,
                    restart_failed=restart, n_fail=nf)
    assert torch.equal(nf, 2 * (~ok).to(torch.int64))
    with pytest.raises(pkg.LmpcError, match="must not alias"):
        sv.loop_advance(trk, dict(inp2, X_ref=out["X_optm"]), out, x2, u2, dt, dt / 2, 2)
    sv.close()


@pytest.mark.parametrize("warm,graph", [(False, False), (True, False), (True, True)])
def test_closed_loop_fused_equals_unfused(pkg, warm, graph):
    B, steps = 512, 60
    tr = pkg.workloads.synthetic_track("barc")
    x0 = torch.as_tensor(_states(tr, B, fast=0.9), dtype=torch.float64, device="cuda")
    u0 = torch.zeros((2, B), dtype=torch.float64, device="cuda")
    res = {}
    for fused in (True, False):
        sv = pkg.Solver(pkg.presets.barc_tracking_mpc(20), pkg.presets.barc_vehicle(), device=0)
        res[fused] = pkg.closed_loop.run(sv, tr, x0, u0, steps=steps, speed_scale=0.9, warm=warm, graph=graph, fused=fused)
        torch.cuda.synchronize()
        sv.close()
    a, b = res[True], res[False]
    # (the last knot's ulp -- see above -- reaches the plant through the next solves: the loops agree to rounding, not bit for bit)
    sx = torch.tensor([2000.0, 10.0, 0.1, 80.0, 2.0, 2.0], dtype=torch.float64, device="cuda")[:, None]
    same = (a["n_fail"] == 0) & (b["n_fail"] == 0)
    err = float((((a["x"] - b["x"]) / sx).abs()[:, same]).max())
    print("fused vs separate launches, %d periods: final states %.1e (scaled), distance %.1e, cars without a failed solve %d of %d"
          % (steps, err, float((a["distance"] - b["distance"]).abs()[same].max()), int(same.sum()), B))
    assert err < 1e-9 and float((a["distance"] - b["distance"]).abs()[same].max()) < 1e-9
    assert float((a["worst_excess"] - b["worst_excess"]).abs()[same].max()) < 1e-9
    assert torch.equal(a["n_fail"], b["n_fail"])
    if warm:
        assert abs(a["warm_hit_rate"] - b["warm_hit_rate"]) < 2e-3
    assert float(a["distance"].median()) > 0.5 * steps * 0.025 * 1.0      # (the cars do move)
