"""Synthetic workloads for the batched LMPC solve path (product side, numpy only).

The BASELINE.json configs are quoted on synthetic data of the reference's scale
(SURVEY.md section 8d): a closed track given as uniform periodic tables
(`lmpc_track` in include/lmpc_hip.h) plus random initial states.  Nothing here
computes on the hot path; it only produces the inputs that are uploaded to HBM.

Track statistics follow the reference's data files
(racing_trajectory/test_data/barc/15_barc_optm.txt: L = 15.63 m, curvature
-0.36 .. 0.96 1/m, speed 2.7 .. 5.2 m/s; putnam/10_putnam_optm.txt: L = 2849 m,
curvature -0.027 .. 0.048 1/m, speed 15 .. 70 m/s) -- the files themselves are
not read or copied.
"""
from __future__ import annotations

import numpy as np


def synthetic_track(kind: str = "barc", M: int = 1024) -> dict:
    """Closed track: curvature integrates to 2*pi over one lap."""
    s = np.arange(M, dtype=np.float64)
    th = 2.0 * np.pi * s / M
    if kind == "barc":
        L = 15.6
        k = (2.0 * np.pi / L) * (1.0 + 0.9 * np.cos(2 * th + 0.3) + 0.45 * np.cos(3 * th - 1.1))
        half_l = 0.55 + 0.25 * np.sin(th + 0.5)
        half_r = 0.55 + 0.25 * np.cos(2 * th - 0.2)
        # speed profile consistent with the grip in the corners (lateral acceleration 0.45 g), as a raceline
        # optimiser would produce; range 1.5 .. 4.5 m/s (the reference's BARC profile spans 2.7 .. 5.2)
        vel = np.clip(np.sqrt(0.45 * 9.8 / np.maximum(np.abs(k), 1e-3)), 1.5, 4.5)
    elif kind == "putnam":
        L = 2849.0
        k = (2.0 * np.pi / L) * (1.0 + 8.0 * np.cos(3 * th + 0.4) ** 3 + 5.0 * np.cos(5 * th - 0.7))
        half_l = 4.5 + 2.5 * np.sin(th + 0.5)
        half_r = 4.5 + 2.5 * np.cos(2 * th - 0.2)
        vel = np.clip(np.sqrt(1.6 * 9.8 / np.maximum(np.abs(k), 1e-4)), 15.0, 65.0)  # 1.6 g with downforce
    else:
        raise ValueError(kind)
    return {"L": L, "M": M, "curvature": k, "bound_left": half_l, "bound_right": -half_r, "vel": vel}


def sample_initial_states(kind: str, batch: int, L: float, u_lo, u_hi, seed: int = 0):
    """Random x_ic (B, 6), u_ic (B, 2) -- SURVEY.md 8d configs 2 (barc) and 4 (putnam)."""
    rng = np.random.default_rng(seed)
    if kind == "barc":
        x = np.stack([rng.uniform(0.0, L, batch), rng.uniform(-0.3, 0.3, batch),
                      rng.normal(0.0, 0.1, batch), rng.uniform(0.5, 3.0, batch),
                      rng.normal(0.0, 0.05, batch), rng.normal(0.0, 0.2, batch)], axis=-1)
        u = np.stack([rng.normal(0.0, 0.002, batch), rng.normal(0.0, 0.1, batch)], axis=-1)
    elif kind == "putnam":
        x = np.stack([rng.uniform(0.0, L, batch), rng.uniform(-1.5, 1.5, batch),
                      rng.normal(0.0, 0.05, batch), rng.uniform(15.0, 70.0, batch),
                      rng.normal(0.0, 0.5, batch), rng.normal(0.0, 0.1, batch)], axis=-1)
        u = np.stack([rng.normal(0.0, 1.0, batch), rng.normal(0.0, 0.02, batch)], axis=-1)
    else:
        raise ValueError(kind)
    u = np.clip(u, np.asarray(u_lo), np.asarray(u_hi))
    return x, u


def synthetic_laps(track: dict, n_laps: int = 5, n_pts: int = 440, v0: float = 1.4):
    """Recorded-lap stand-ins for the LMPC safe set: n_pts samples per lap
    along the track, a small lateral weave that differs per lap, speed rising from lap to lap
    (statistics of the reference's barc_ss laps: ~440 samples, v ~ 1.5 m/s, |e_y| < 0.1)."""
    L, M = float(track["L"]), int(track["M"])
    laps = []
    for l in range(n_laps):
        s = (np.arange(n_pts) + 0.37) * L / n_pts
        k = np.interp(s, np.arange(M) * L / M, track["curvature"], period=L)
        vx = np.full(n_pts, v0 + 0.05 * l)
        ey = 0.06 * np.sin(2 * np.pi * 3 * s / L + 0.9 * l)
        epsi = 0.06 * (2 * np.pi * 3 / L) * np.cos(2 * np.pi * 3 * s / L + 0.9 * l)
        laps.append(np.stack([s, ey, epsi, vx, np.zeros(n_pts), k * vx], axis=1))
    return laps


def sample_states_near_laps(laps, batch: int, L: float, seed: int = 0):
    rng = np.random.default_rng(seed)
    lap = laps[-1]
    idx = rng.integers(0, lap.shape[0], batch)
    x = lap[idx] + rng.normal(0, 1, (batch, 6)) * np.array([0.0, 0.03, 0.03, 0.1, 0.02, 0.1])
    x[:, 0] = np.mod(x[:, 0], L)
    return x, np.zeros((batch, 2))


def regression_sample_pairs(track: dict, laps, plant_step, seed: int = 7, dt: float = 0.03):
    """Recorded data of a plant that differs from the model, as two-sample laps for `Solver.set_regression_laps`
    (BASELINE configs[4]: "LMPC + error-dynamics residual term"): states around the stored laps, inputs drawn at random,
    and each state's successor one `dt` later from `plant_step(xa [n, 6], ua [n, 2]) -> xb [n, 6]` -- the caller's plant
    (bench.py and the full-size parity test use the plant kernel of a second handle with 15 % less grip)."""
    rng = np.random.default_rng(seed)
    n = sum(l.shape[0] for l in laps)
    xa = np.concatenate(laps) + rng.normal(0, 1, (n, 6)) * np.array([0.0, 0.02, 0.02, 0.1, 0.03, 0.2])
    ua = np.stack([rng.uniform(-0.005, 0.005, n), rng.uniform(-0.15, 0.15, n)], axis=1)
    ka = np.interp(xa[:, 0], np.arange(track["M"]) * track["L"] / track["M"], track["curvature"], period=track["L"])
    xb = np.asarray(plant_step(xa, ua))
    return [(np.stack([xa[j], xb[j]]), np.stack([ua[j], ua[j]]), np.array([ka[j], ka[j]]), np.array([0.0, dt])) for j in range(n)]


def track_from_file(path, M: int = 1024) -> dict:
    """Uniform periodic device tables sampled from one of the reference's 17-column track files
    (racing_trajectory.py: same interpolants as RacingTrajectory upstream)."""
    from .racing_trajectory import RacingTrajectory

    return RacingTrajectory(path).to_track_table(M)
