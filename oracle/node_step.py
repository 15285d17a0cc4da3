"""One tick of the controller node, restated (oracle; test infrastructure -- only tests/ may import this).

Follows RacingMPCNode::on_step_timer, src/mpc/racing_mpc/src/racing_mpc_node.cpp:
  :181-185  the state message's global pose projected to the Frenet frame (the projection itself is checked by the round
            trip frenet_to_global(result) == pose: the reference solves it with a CasADi rootfinder, a third-party
            dependency that is not vendored, so this file takes the Frenet pose as an input);
  :191-202  from_base_control of the last actuation message (single_track_planar_model.cpp:399-405: the larger in
            magnitude of (u_a)+ and (u_a)-, and the steering);
  :210-235  first call: zero-input rollout U = 1e-9 from x_ic with the model's discrete dynamics, curvature taken at each
            knot's own abscissa (:68-76);
  :236-259  later calls: x_ic one model step ahead with the input about to be applied (CONTINUOUS) or as measured (STEP);
            the previous plan shifted by one knot, the last input repeated, the last input rate zero, the last state
            rolled out from the one before it;
  :261-292  boundaries, curvature and velocity reference at the plan's abscissae; the velocity reference scaled by
            speed_scale, clamped to current speed +- max_vel_ref_diff and capped by the (equally clamped) speed limit; a
            non-positive profile means "use the speed limit";
  :385-402  the actuation message from column delay_step of the plan through to_base_control
            (single_track_planar_model.cpp:393-398).
"""
from __future__ import annotations

import numpy as np

from .dynamics import rk4
from .params import Vehicle
from .trajectory import TrackOracle


def from_base_control(u_a: float, u_steer: float) -> np.ndarray:
    fd, fb = (u_a if u_a > 0.0 else 0.0), (u_a if u_a < 0.0 else 0.0)   # racing_mpc_node.cpp:191-195
    return np.array([fd if abs(fd) > abs(fb) else fb, u_steer])          # single_track_planar_model.cpp:399-405


def to_base_control(u: np.ndarray) -> np.ndarray:
    return np.array([u[0] / (1.0 + np.exp(-u[0])), u[0] / (1.0 + np.exp(u[0])), u[1]])  # :393-398


def actuation(U: np.ndarray, delay_step: int = 0):
    """(u_a, u_steer) published from the plan U (2, N-1): racing_mpc_node.cpp:385-402."""
    ub = to_base_control(U[:, delay_step])
    return (ub[0] if abs(ub[0]) > abs(ub[1]) else ub[1]), ub[2]


def _step_model(track: TrackOracle, veh: Vehicle, x, u, dt):
    return rk4(np.asarray(x, float), np.asarray(u, float), track.eval(x[0])["curvature"], dt, veh)  # :68-76


def step_inputs(track: TrackOracle, veh: Vehicle, N: int, dt: float, max_vel_ref_diff: float, x_frenet, act_in, last,
                *, continuous: bool = True, speed_limit: float = np.inf, speed_scale: float = 1.0) -> dict:
    """sol_in of one tick.  x_frenet: the measured state (6,) in the Frenet frame; act_in: (u_a, u_steer) of the last
    actuation message; last: None on the first call, else (X (6, N), U (2, N-1), dU (2, N-1)) of the previous plan."""
    x_ic = np.asarray(x_frenet, float)
    out = {"u_ic": from_base_control(*act_in), "T_ref": np.full(N - 1, dt)}
    if last is None:
        X, U, dU = np.zeros((6, N)), np.full((2, N - 1), 1e-9), np.zeros((2, N - 1))
        X[:, 0] = x_ic
        for i in range(1, N):
            X[:, i] = _step_model(track, veh, X[:, i - 1], U[:, i - 1], dt)
        out["x_ic"] = x_ic
    else:
        Xp, Up, dUp = (np.asarray(a, float) for a in last)
        out["x_ic"] = _step_model(track, veh, x_ic, Up[:, 0], dt) if continuous else x_ic
        X = np.concatenate([Xp[:, 1:], np.zeros((6, 1))], axis=1)
        U = np.concatenate([Up[:, 1:], Up[:, -1:]], axis=1)
        dU = np.concatenate([dUp[:, 1:], np.zeros((2, 1))], axis=1)
        X[:, -1] = _step_model(track, veh, X[:, -2], U[:, -1], dt)
    out.update(X_ref=X, U_ref=U, X_optm_ref=X, U_optm_ref=U, dU_optm_ref=dU)
    ref = track.eval(X[0])
    vel = np.empty(N)
    for i in range(N):
        cur, want = X[3, i], ref["vel"][i] * speed_scale
        cap = np.clip(speed_limit, cur - max_vel_ref_diff, cur + max_vel_ref_diff)
        vel[i] = min(np.clip(want, cur - max_vel_ref_diff, cur + max_vel_ref_diff), cap) if want > 0.0 else cap
    out.update(bound_left=ref["left"], bound_right=ref["right"], curvatures=ref["curvature"], vel_ref=vel)
    return out
