"""Closed-loop Monte-Carlo harness: thousands of cars driven by the batched MPC, entirely on the device.

One control period per iteration, as the reference's two nodes do it (step mode):
  RacingMPCNode::on_step_timer   shift previous plan, re-sample references, solve   (racing_mpc_node.cpp:236-320)
  RacingSimulatorNode / ::step   plant RK4 with the first input of the plan          (racing_simulator.cpp:97-112)
Everything (prepare / shift / linearise / solve / plant) is a HIP kernel behind the C ABI; this module only
sequences the calls and keeps a few statistics.  SURVEY.md 8d config 1 is this loop with batch = 1.
"""
from __future__ import annotations


def run(solver, track: dict, x0, u0, steps: int, dt: float = 0.025, n_sub: int = 2, speed_scale: float = 0.9,
        record_every: int = 0, restart_failed: bool = True, graph: bool = False, longest_first: bool = False, warm: bool = False,
        warm_rounds: int = 0, fused: bool = True):
    """x0 [6][B], u0 [2][B] (torch, device).  Returns final state and statistics (torch tensors on device).

    One control period = solve, apply the first input of the plan (on failure: of the shifted previous plan,
    racing_mpc_node.cpp:322-332), plant step, statistics, warm-start shift; a car whose QP failed is re-prepared from a
    cold start at its current state, as re-launching the node would (racing_mpc_node.cpp:210-235) -- a stale plan would
    make the next QP fail too.  Nothing in a period reads device data on the host.

    graph=True captures the period once as a HIP graph and replays it: a period is ~25 small launches around the QP
    kernel, and eager dispatch (~2 ms of host time) costs more than the 0.9 ms the GPU needs for them.

    warm=True solves with lmpc_solve_batch_warm: the shifted previous plan (what `inp["X_ref"]`, `inp["U_ref"]` hold from the second
    period on, racing_mpc_node.cpp:245-254) is tried as an active-set solve before any interior point.  Returns the share of
    solves that took that route as "warm_hit_rate" (iters <= 4: an attempt has four rounds at most and a refused one reports its
    rounds + the cold solve's iterations, five at least).  warm_rounds: lmpc_set_warm_rounds for this run (0: the library's default
    by batch size); the handle is back on the default afterwards.

    fused=True (default) does everything behind the solve -- input selection, plant step, statistics, shift or cold restart -- with
    one launch, lmpc_loop_advance_batch; fused=False goes through lmpc_plant_step_batch, lmpc_shift_batch, lmpc_prepare_failed_batch
    and torch element-wise operations (~45 launches per period), the same arithmetic (tests/test_gpu_loop.py: bit for bit but for 1 - 2 ulp on the last knot's rollout).

    longest_first=True launches the QP kernel's workgroups in the order of the previous period's iteration counts, longest
    first (lmpc_set_launch_order): a car's count changes little from one period to the next, and the long problems then
    no longer start last in the second residency round."""
    import torch

    trk = solver.device_track(track)
    L = float(track["L"])
    x = x0.clone()
    u_prev = u0.clone()
    B = x.shape[1]
    inp = solver.prepare(trk, x, dt, speed_scale=speed_scale)          # cold start (racing_mpc_node.cpp:210-235)
    out = solver.alloc_outputs(B)
    dist = torch.zeros(B, dtype=torch.float64, device=x.device)         # abscissa travelled (unwrapped)
    worst_excess = torch.zeros(B, dtype=torch.float64, device=x.device)  # max lateral excursion beyond the track edge
    n_fail = torch.zeros(B, dtype=torch.int64, device=x.device)
    hits = torch.zeros((), dtype=torch.int64, device=x.device)
    half_b = float(solver.vehicle["b"]) / 2.0
    keys = ("X_ref", "U_ref", "T_ref", "bound_left", "bound_right", "curvatures", "vel_ref")
    trace = []
    if warm:
        solver.set_warm_rounds(warm_rounds)
    order = None
    if longest_first:
        order = torch.arange(B, dtype=torch.int32, device=x.device)
        solver.set_launch_order(order)

    def period():
        inp["x_ic"] = x
        inp["u_ic"] = u_prev
        solver.solve(inp, out, warm=True if warm else None)
        if order is not None:
            solver.launch_order_from_iters(out["iters"], order)   # for the next period (in place: the pointer is registered)
        if fused:
            solver.loop_advance(trk, inp, out, x, u_prev, dt, dt / n_sub, n_sub, speed_scale=speed_scale, restart_failed=restart_failed,
                                distance=dist, worst_excess=worst_excess, n_fail=n_fail, n_accepted=hits if warm else None)
            return
        if warm:
            hits.add_(((out["status"] == 0) & (out["iters"] <= 4)).sum())   # (an attempt has 4 rounds at most; a cold solve takes 5 iterations at least)
        ok = out["status"] == 0
        n_fail.add_((~ok).to(torch.int64))
        u_apply = torch.where(ok[None, :], out["U_optm"][:, 0, :], inp["U_ref"][:, 0, :]).contiguous()
        s_before = x[0].clone()
        solver.plant_step(trk, x, u_apply, dt / n_sub, n_sub)
        ds = x[0] - s_before
        dist.add_(torch.where(ds < -L / 2, ds + L, ds))
        exc = torch.maximum(x[1] + half_b - inp["bound_left"][0], inp["bound_right"][0] - (x[1] - half_b))
        torch.maximum(worst_excess, exc, out=worst_excess)
        u_prev.copy_(u_apply)
        nxt = solver.shift(trk, inp, out, dt, speed_scale=speed_scale)
        if restart_failed:
            solver.prepare_failed(trk, x, out["status"], nxt, dt, speed_scale=speed_scale)
        for key in keys:
            inp[key].copy_(nxt[key])

    if not graph:
        for k in range(steps):
            period()
            if record_every and k % record_every == 0:
                trace.append(x.clone())
    else:
        if record_every:
            raise ValueError("record_every is not available with graph=True")
        done = 0
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side):           # warm-up outside the capture: workspace allocation, lazy initialisation
            if steps > 0:
                period()
                done = 1
        torch.cuda.current_stream(x.device).wait_stream(side)
        if steps > done:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                period()
            done += 1                               # (the capture does not execute; the first replay is that period)
            g.replay()
            for _ in range(steps - done):
                g.replay()
    if order is not None:
        torch.cuda.synchronize(x.device)
        solver.set_launch_order(None)
    if warm and warm_rounds:
        solver.set_warm_rounds(0)
    return {"x": x, "distance": dist, "worst_excess": worst_excess, "n_fail": n_fail, "trace": trace,
            "warm_hit_rate": (float(hits) / (B * max(steps, 1))) if warm else None}


def record_laps(solver, track: dict, speed_scales=(0.80, 0.85, 0.90, 0.95, 1.0), dt: float = 0.03, n_sub: int = 3):
    """The laps SURVEY.md 8d config 3 stores in the safe set: "running config 1's tracking loop for 5 laps with seed-indexed speed
    scales {0.80, 0.85, 0.90, 0.95, 1.0}", one sample per 0.03 s (the recorder's period upstream, racing_mpc_node.cpp:66).  One
    car per scale from the start line at the profile's speed, driven by `solver` (a tracking handle) until it has covered one lap;
    the samples of that lap are returned as host arrays [n][6], oldest (slowest) first -- what SafeSetRecorder would have written."""
    import numpy as np
    import torch

    L, M = float(track["L"]), int(track["M"])
    laps = []
    for sc in speed_scales:
        v0 = max(float(np.asarray(track["vel"])[0]) * sc, 0.5)
        x0 = torch.tensor([[0.0], [0.0], [0.0], [v0], [0.0], [0.0]], dtype=torch.float64, device=solver.device)
        u0 = torch.zeros((2, 1), dtype=torch.float64, device=solver.device)
        v_min = max(float(np.asarray(track["vel"]).min()) * sc * 0.8, 0.4)
        steps = int(1.25 * L / (v_min * dt)) + 8
        r = run(solver, track, x0, u0, steps=steps, dt=dt, n_sub=n_sub, speed_scale=sc, record_every=1)
        xs = np.stack([x0.cpu().numpy()[:, 0]] + [t.cpu().numpy()[:, 0] for t in r["trace"]])     # [steps + 1][6], abscissa wrapped
        ds = np.diff(xs[:, 0])
        ds = np.where(ds < -L / 2, ds + L, ds)
        travelled = np.concatenate([[0.0], np.cumsum(ds)])
        n = int(np.searchsorted(travelled, L))       # first sample past the line: the lap is samples 0 .. n-1
        if n >= xs.shape[0]:
            raise RuntimeError("record_laps: the car at speed scale %.2f did not complete a lap in %d periods" % (sc, steps))
        laps.append(xs[:n].copy())
    return laps


def run_lmpc(tracker, learner, track: dict, x0, u0, warm_laps: int = 2, learn_laps: int = 4, dt: float = 0.025,
             n_sub: int = 2, warm_speed_scale: float = 0.7, max_steps: int = 20000, debug: bool = False, warm: bool = False,
             advance: int = 1):
    """The LMPC experiment of the reference (sim_barc_lmpc): `warm_laps` laps under the tracking MPC fill the safe
    set, then the learning MPC drives and every completed lap of car 0 is added to the set (SafeSetRecorder ->
    SafeSetManager -> device store).  All B cars share car 0's safe set.  Returns car 0's lap times (tracking laps
    first) and per-car statistics.  x0 [6][B], u0 [2][B] on the device.

    warm=True (round 6): the learning solves go through lmpc_solve_batch_warm_ss -- the shifted previous plan and the previous
    solution's simplex weights (racing_mpc.cpp:281, 293-305), the safe set by reference, the weights carried onto the new period's
    points by lmpc_shift_lambda_batch (`advance` samples along the lap); the tracking laps through lmpc_solve_batch_warm.  Returns
    the share of the learning solves whose active-set attempt was accepted as "warm_hit_rate"."""
    import numpy as np
    import torch

    from . import safe_set as SS

    trk = tracker.device_track(track)
    L = float(track["L"])
    B = x0.shape[1]
    man = SS.SafeSetManager(int(learner.config["max_lap_stored"]))
    rec = SS.SafeSetRecorder(man)
    x, u_prev = x0.clone(), u0.clone()
    half_b = float(tracker.vehicle["b"]) / 2.0
    worst_excess = torch.zeros(B, dtype=torch.float64, device=x.device)
    n_fail = torch.zeros(B, dtype=torch.int64, device=x.device)
    lap_times, lap_kind, t_lap_start, t = [], [], None, 0.0
    debug_left = [12]
    solver, learning = tracker, False
    inp = solver.prepare(trk, x, dt, speed_scale=warm_speed_scale)
    out = solver.alloc_outputs(B)
    S = int(learner.config["num_ss_pts"])
    lam = torch.zeros((S, B), dtype=torch.float64, device=x.device)
    curv0 = np.asarray(track["curvature"], dtype=np.float64)
    # The state box also applies to knot 0 (racing_mpc.cpp:147,201), so a plant that lands a hair outside an active
    # bound (the QP rides vx = vx_max; the nonlinear plant overshoots by the linearisation error) would make every
    # following problem infeasible.  The harness hands the controller the measured state projected onto the box.
    x_lo = torch.as_tensor(learner.config["x_min"], dtype=torch.float64, device=x.device)[:, None]
    x_hi = torch.as_tensor(learner.config["x_max"], dtype=torch.float64, device=x.device)[:, None]
    idx_prev, have_prev, n_warm, n_hit = None, False, 0, 0
    acc = torch.zeros(B, dtype=torch.int32, device=x.device)
    for k in range(max_steps):
        inp["x_ic"], inp["u_ic"] = (torch.minimum(torch.maximum(x, x_lo), x_hi) if learning else x), u_prev
        # car 0 feeds the recorder (RacingMPC::solve, racing_mpc.cpp:246): state, applied input, curvature, time
        x_h = x[:, 0].cpu().numpy()
        k_h = float(np.interp(x_h[0] % L, np.arange(curv0.size) * L / curv0.size, curv0, period=L))
        if rec.step(x_h, u_prev[:, 0].cpu().numpy(), k_h, t, L):
            lap_times.append(t - t_lap_start)
            lap_kind.append("lmpc" if learning else "tracking")
            man.sync(learner)
            have_prev = False   # (the store was replaced: the previous period's codes no longer name its rows)
            if not learning and len(man.laps) >= warm_laps:
                solver, learning = learner, True
                out = solver.alloc_outputs(B)
                out["convex_combi_optm"] = lam
            if learning and lap_kind.count("lmpc") >= learn_laps:
                break
        if rec.initialized and (t_lap_start is None or rec.x and len(rec.x) == 1):
            t_lap_start = t
        if learning:
            # query = last knot of the abscissa-aligned reference (racing_mpc.cpp:219-223,249-254)
            s_last, s0 = inp["X_ref"][0, -1], x[0]
            kk = (s0 - s_last).abs() + L / 2
            q = torch.stack([s_last + (kk - torch.fmod(kk, L)) * torch.sign(s0 - s_last), inp["X_ref"][1, -1]]).contiguous()
            if warm:
                idx, _ = solver.ss_query_idx(q)
                idx = idx.clone()  # (kept for the next period's weight transfer)
                if have_prev:
                    lam_ref = solver.shift_lambda(idx_prev, lam, idx, advance)
                    solver.solve(inp, out, ss_idx=idx, warm={"X_optm_ref": inp["X_ref"], "U_optm_ref": inp["U_ref"], "convex_combi_optm_ref": lam_ref})
                    solver.warm_accepted(B, acc)
                    n_warm += B
                    n_hit += int(acc.sum())
                else:
                    solver.solve(inp, out, ss_idx=idx)
                idx_prev = idx
            else:
                ss_x, ss_j, _ = solver.ss_query(q)
                solver.solve(inp, out, ss_x=ss_x, ss_j=ss_j)
        else:
            solver.solve(inp, out, warm=True if (warm and k > 0) else None)
        ok = out["status"] == 0
        n_fail += (~ok).to(torch.int64)
        if learning and warm:
            have_prev = bool(ok.all()) or True   # (a failed car's weights are stale; its attempt is refused by the KKT test, nothing else)
        if debug and learning and not bool(ok[0]) and debug_left[0] > 0:
            debug_left[0] -= 1
            print("step", k, "t %.3f" % t, "car0 status", int(out["status"][0]), "iters", int(out["iters"][0]), "x", x[:, 0].cpu().numpy().round(3),
                  "u_prev", u_prev[:, 0].cpu().numpy().round(4), "Xref vx %.2f..%.2f" % (float(inp["X_ref"][3, :, 0].min()), float(inp["X_ref"][3, :, 0].max())))
        u_apply = torch.where(ok[None, :], out["U_optm"][:, 0, :], inp["U_ref"][:, 0, :]).contiguous()
        solver.plant_step(trk, x, u_apply, dt / n_sub, n_sub)
        exc = torch.maximum(x[1] + half_b - inp["bound_left"][0], inp["bound_right"][0] - (x[1] - half_b))
        worst_excess = torch.maximum(worst_excess, exc)
        u_prev = u_apply
        inp = solver.shift(trk, inp, out, dt, speed_scale=warm_speed_scale if not learning else 1.0)
        t += dt
    return {"lap_times": lap_times, "lap_kind": lap_kind, "worst_excess": worst_excess, "n_fail": n_fail, "steps": k + 1,
            "laps_in_set": len(man.laps), "x": x, "warm_hit_rate": (n_hit / n_warm) if n_warm else None}
