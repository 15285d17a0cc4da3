"""Input preparation as the reference's controller node does it (oracle; test infrastructure).

Restates RacingMPCNode::on_step_timer's cold-start branch and reference sampling
(src/mpc/racing_mpc/src/racing_mpc_node.cpp:210-235, 261-292) on a track given
as uniform periodic tables (the product's own track format, see
include/lmpc_hip.h `lmpc_track`; the reference interpolates B-splines instead,
racing_trajectory.cpp:25-120 -- out of scope, the tables are inputs here).
"""
from __future__ import annotations

import numpy as np

from . import dynamics as dyn
from .params import MPCConfig, Vehicle


def track_lookup(tab: np.ndarray, s, L: float):
    """Periodic linear interpolation on a uniform grid of M samples over [0, L)."""
    M = tab.shape[0]
    u = np.mod(np.asarray(s, dtype=np.float64), L) / (L / M)
    i0 = np.floor(u).astype(np.int64)
    fr = u - i0
    i0 = np.mod(i0, M)
    i1 = np.mod(i0 + 1, M)
    return tab[i0] * (1.0 - fr) + tab[i1] * fr


def cold_start_inputs(cfg: MPCConfig, veh: Vehicle, track: dict, x_ic, u_ic, dt: float,
                      speed_scale: float = 1.0, speed_limit: float | None = None) -> dict:
    """Batch of solver inputs.  x_ic: (B, 6), u_ic: (B, 2).  Returns arrays with the
    batch axis LAST (the C-ABI's [field][time][batch] layout)."""
    x_ic = np.atleast_2d(np.asarray(x_ic, dtype=np.float64))
    u_ic = np.atleast_2d(np.asarray(u_ic, dtype=np.float64))
    B, N, L = x_ic.shape[0], cfg.N, float(track["L"])
    if speed_limit is None:
        speed_limit = float(cfg.x_max[3])  # racing_mpc_node.hpp:69
    X = np.zeros((N, B, 6))
    U = np.full((N - 1, B, 2), 1e-9)  # :212
    X[0] = x_ic
    for i in range(1, N):  # :216-224, curvature looked up at the knot's own abscissa (:72-76)
        k = track_lookup(track["curvature"], X[i - 1][:, 0], L)
        X[i] = dyn.rk4(X[i - 1], U[i - 1], k, dt, veh)
    s = X[:, :, 0]
    bl = track_lookup(track["bound_left"], s, L)
    br = track_lookup(track["bound_right"], s, L)
    kap = track_lookup(track["curvature"], s, L)
    vr = track_lookup(track["vel"], s, L) * speed_scale
    cur = X[:, :, 3]
    d = cfg.max_vel_ref_diff
    lim = np.clip(speed_limit, cur - d, cur + d)  # :273-275
    vref = np.where(vr > 0.0, np.minimum(np.clip(vr, cur - d, cur + d), lim), lim)  # :276-285
    return {
        "x_ic": x_ic.T.copy(), "u_ic": u_ic.T.copy(),
        "X_ref": np.ascontiguousarray(X.transpose(2, 0, 1)),       # [6][N][B]
        "U_ref": np.ascontiguousarray(U.transpose(2, 0, 1)),       # [2][N-1][B]
        "T_ref": np.full((N - 1, B), dt),
        "bound_left": bl, "bound_right": br, "curvatures": kap, "vel_ref": vref,
        "L": L,
    }


def problem(inputs: dict, b: int) -> dict:
    """Slice problem b out of a batch into the per-problem dict oracle/qp.py consumes."""
    return {
        "x_ic": inputs["x_ic"][:, b], "u_ic": inputs["u_ic"][:, b],
        "X_ref": inputs["X_ref"][:, :, b], "U_ref": inputs["U_ref"][:, :, b],
        "T_ref": inputs["T_ref"][:, b],
        "bound_left": inputs["bound_left"][:, b], "bound_right": inputs["bound_right"][:, b],
        "curvatures": inputs["curvatures"][:, b], "vel_ref": inputs["vel_ref"][:, b],
        "L": inputs["L"],
    }


def _sample_refs(cfg, track, X, speed_scale, speed_limit):
    L = float(track["L"])
    s = X[:, :, 0]
    bl = track_lookup(track["bound_left"], s, L)
    br = track_lookup(track["bound_right"], s, L)
    kap = track_lookup(track["curvature"], s, L)
    vr = track_lookup(track["vel"], s, L) * speed_scale
    cur = X[:, :, 3]
    d = cfg.max_vel_ref_diff
    lim = np.clip(speed_limit, cur - d, cur + d)
    vref = np.where(vr > 0.0, np.minimum(np.clip(vr, cur - d, cur + d), lim), lim)
    return bl, br, kap, vref


def shift_inputs(cfg: MPCConfig, veh: Vehicle, track: dict, X_prev, U_prev, dt: float,
                 speed_scale: float = 1.0, speed_limit: float | None = None) -> dict:
    """racing_mpc_node.cpp:245-254: shift the previous plan one knot, repeat the last input, roll out the last
    state, re-sample the references (:261-292).  X_prev [6][N][B], U_prev [2][N-1][B]."""
    N, L = cfg.N, float(track["L"])
    if speed_limit is None:
        speed_limit = float(cfg.x_max[3])
    X = np.empty_like(X_prev)
    U = np.empty_like(U_prev)
    X[:, :N - 1] = X_prev[:, 1:]
    U[:, :N - 2] = U_prev[:, 1:]
    U[:, N - 2] = U_prev[:, N - 2]
    k_last = track_lookup(track["curvature"], X[0, N - 2], L)
    X[:, N - 1] = dyn.rk4(X[:, N - 2].T, U[:, N - 2].T, k_last, dt, veh).T
    bl, br, kap, vref = _sample_refs(cfg, track, X.transpose(1, 2, 0), speed_scale, speed_limit)
    return {"X_ref": X, "U_ref": U, "T_ref": np.full((N - 1, X.shape[2]), dt), "bound_left": bl,
            "bound_right": br, "curvatures": kap, "vel_ref": vref, "L": L}


def plant_step(veh: Vehicle, track: dict, x, u, dt_sim: float, n_sub: int = 1):
    """racing_simulator.cpp:97-112 with the abscissa wrap of :61-64.  x (B, 6), u (B, 2)."""
    L = float(track["L"])
    x = np.array(x, dtype=np.float64)
    for _ in range(n_sub):
        small = np.abs(x[:, 3]) < 1e-6
        x[small, 3] = np.copysign(1e-6, x[small, 3])
        k = track_lookup(track["curvature"], x[:, 0], L)
        x = dyn.rk4(x, u, k, dt_sim, veh)
        x[:, 0] = dyn.align_abscissa(x[:, 0], L / 2.0, L)
    return x
