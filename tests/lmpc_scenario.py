"""BARC LMPC scenario built on the reference's recorded laps (tests/golden/barc_ss): shared by the golden
generator and the tests.  The track tables take lap 1's recorded curvature; bounds and speed are flat."""
from pathlib import Path

import numpy as np

from oracle import cbind, dynamics as D, params as P, scenario as S

GOLD = Path(__file__).resolve().parent / "golden" / "barc_ss"
L_BARC_SS = 17.06  # lap length of the recorded laps (s wraps at ~17.02 + one sample)


def load_laps():
    return [np.loadtxt(GOLD / f"ss_lap_{i}_x.txt") for i in (1, 2, 3)]


def track_from_lap(M: int = 512) -> dict:
    lap = np.loadtxt(GOLD / "ss_lap_1_x.txt")
    k = np.loadtxt(GOLD / "ss_lap_1_k.txt")
    order = np.argsort(lap[:, 0])
    sg = np.arange(M) * L_BARC_SS / M
    curv = np.interp(sg, lap[order, 0], k[order], period=L_BARC_SS)
    return {"L": L_BARC_SS, "M": M, "curvature": curv, "bound_left": np.full(M, 0.55),
            "bound_right": np.full(M, -0.55), "vel": np.full(M, 2.0)}


def make(B: int, seed: int, N: int = 20, n_laps: int = 3):
    """Inputs of a batch of LMPC problems: states near lap 3, cold-start references, query points."""
    laps = load_laps()
    veh, cfg = P.barc_vehicle(), P.barc_lmpc(N, n_laps)
    tr = track_from_lap()
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, laps[2].shape[0], B)
    x = laps[2][idx] + rng.normal(0, 1, (B, 6)) * np.array([0.0, 0.03, 0.03, 0.1, 0.02, 0.1])
    x[:, 0] = np.mod(x[:, 0], L_BARC_SS)
    inp = S.cold_start_inputs(cfg, veh, tr, x, np.zeros((B, 2)), 0.025)
    # racing_mpc.cpp:219-223,249-254: the query is the last knot of the abscissa-aligned reference
    q = np.stack([D.align_abscissa(inp["X_ref"][0, -1, :], inp["x_ic"][0, :], L_BARC_SS), inp["X_ref"][1, -1, :]])
    return veh, cfg, tr, laps, inp, q


def oracle_safe_set(cfg, laps, q):
    return cbind.ss_query_batch(laps[-cfg.max_lap_stored:], L_BARC_SS, cfg.num_ss_pts, cfg.num_ss_pts_per_lap, q)
