"""Closed-loop Monte-Carlo harness: thousands of cars driven by the batched MPC, entirely on the device.

One control period per iteration, as the reference's two nodes do it (step mode):
  RacingMPCNode::on_step_timer   shift previous plan, re-sample references, solve   (racing_mpc_node.cpp:236-320)
  RacingSimulatorNode / ::step   plant RK4 with the first input of the plan          (racing_simulator.cpp:97-112)
Everything (prepare / shift / linearise / solve / plant) is a HIP kernel behind the C ABI; this module only
sequences the calls and keeps a few statistics.  SURVEY.md 8d config 1 is this loop with batch = 1.
"""
from __future__ import annotations


def run(solver, track: dict, x0, u0, steps: int, dt: float = 0.025, n_sub: int = 2, speed_scale: float = 0.9,
        record_every: int = 0):
    """x0 [6][B], u0 [2][B] (torch, device).  Returns final state and statistics (torch tensors on device)."""
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
    half_b = float(solver.vehicle["b"]) / 2.0
    trace = []
    for k in range(steps):
        inp["x_ic"] = x
        inp["u_ic"] = u_prev
        solver.solve(inp, out)
        ok = out["status"] == 0
        n_fail += (~ok).to(torch.int64)
        # first input of the plan; on failure the shifted previous plan's first input (racing_mpc_node.cpp:322-332)
        u_apply = torch.where(ok[None, :], out["U_optm"][:, 0, :], inp["U_ref"][:, 0, :]).contiguous()
        s_before = x[0].clone()
        solver.plant_step(trk, x, u_apply, dt / n_sub, n_sub)
        ds = x[0] - s_before
        dist += torch.where(ds < -L / 2, ds + L, ds)
        bl = inp["bound_left"][0]
        br = inp["bound_right"][0]
        exc = torch.maximum(x[1] + half_b - bl, br - (x[1] - half_b))
        worst_excess = torch.maximum(worst_excess, exc)
        u_prev = u_apply
        inp = solver.shift(trk, inp, out, dt, speed_scale=speed_scale)
        if record_every and k % record_every == 0:
            trace.append(x.clone())
    return {"x": x, "distance": dist, "worst_excess": worst_excess, "n_fail": n_fail, "trace": trace}
