"""The problem sets behind tests/golden/dense_*.npz (>= 512 dense optima per BASELINE config, compared with
the KERNEL directly).  Shared by the generator (tests/golden/make_dense_fixtures.py), the CPU test (twin against the fixtures,
tests/test_dense_fixtures.py) and the GPU test (kernel against the fixtures, tests/test_gpu_dense_fixtures.py).

A fixture stores what is expensive -- the dense, polished, KKT-certified optimum of every problem (oracle/qp.py) -- and a digest
of the inputs; the inputs themselves are rebuilt here, on the CPU, from a seed: the product's synthetic workload generator
(numpy) + the oracle's restatement of the node's cold start (oracle/scenario.py) + the oracle's safe-set query.  They are the
FIRST `count` problems of the 4096-batch bench.py draws for the same configuration, so the fixtures pin the distribution the
headline is measured on."""
from __future__ import annotations

import numpy as np

from oracle import cbind, params as P, scenario as S

BATCH = 4096   # the draw the first `count` problems are taken from (bench.py's batch)
INPUT_KEYS = ("x_ic", "u_ic", "X_ref", "U_ref", "T_ref", "bound_left", "bound_right", "curvatures", "vel_ref")

# name: (family, N, count, BASELINE config it pins)
CASES = {
    "barc_tracking_n20": ("trk", 20, 512, "configs[1]"),
    "iac_tracking_n40": ("iac", 40, 512, "configs[3]"),
    "barc_lmpc_n20_s160": ("lrn", 20, 512, "configs[2] / configs[4] (5 laps, 160 points)"),
    "barc_lmpc_n20_s96": ("rec", 20, 512, "configs[2] on the reference's recorded laps (3 laps, 96 points)"),
    "barc_lmpc_spec_n20_s160": ("spc", 20, 512, "configs[2] / configs[4] AS bench.py QUOTES THEM since round 5 (--lmpc-data spec): laps recorded by the "
                                               "tracking loop at five speed scales, configs[1]'s random x0 (SURVEY.md 8d config 3)"),
    "barc_tracking_n40": ("trk", 40, 128, "shipped horizon"),
    "barc_tracking_n60": ("trk", 60, 96, "shipped horizon (barc_tracking_mpc.param.yaml)"),
    "barc_tracking_n80": ("trk", 80, 64, "shipped horizon"),
    "iac_tracking_n80": ("iac", 80, 64, "shipped horizon (iac_car_tracking_mpc.param.yaml)"),
    "barc_lmpc_n40_s160": ("lrn", 40, 96, "shipped horizon (barc_lmpc.param.yaml)"),
    "barc_lmpc_n60_s160": ("lrn", 60, 64, "shipped horizon (iac_car_lmpc.param.yaml's N)"),
}


def spec_laps():
    """The committed recording of SURVEY.md 8(d) config 3's laps: [n][6] each, oldest (slowest) first."""
    from pathlib import Path

    with np.load(Path(__file__).resolve().parent / "golden" / "spec_laps.npz") as z:
        return [z["lap%d" % i] for i in range(5)]


def ss_query_point(inp: dict, L: float) -> np.ndarray:
    """(s, e_y) of the last reference knot, abscissa aligned to x_ic (racing_mpc.cpp:219-223, 249-254; lmpc_utils/utils.hpp:35-41) --
    as bench.py forms the safe-set query."""
    s_last, s0 = inp["X_ref"][0, -1], inp["x_ic"][0]
    kk = np.abs(s0 - s_last) + L / 2
    return np.stack([s_last + (kk - np.fmod(kk, L)) * np.sign(s0 - s_last), inp["X_ref"][1, -1]])


def build(pkg, name: str):
    """-> cfg, veh, inp (batch axis last, `count` problems), ss_x, ss_j (None for tracking)"""
    family, N, count, _ = CASES[name]
    wl = pkg.workloads
    if family == "trk":
        tr = wl.synthetic_track("barc")
        cfg, veh = P.barc_tracking_mpc(N), P.barc_vehicle()
        x, u = wl.sample_initial_states("barc", BATCH, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], seed=0)
        return cfg, veh, S.cold_start_inputs(cfg, veh, tr, x[:count], u[:count], 0.025), None, None
    if family == "iac":
        tr = wl.synthetic_track("putnam")
        cfg, veh = P.iac_tracking_mpc(N), P.iac_vehicle()
        x, u = wl.sample_initial_states("putnam", BATCH, tr["L"], [-10.0, -0.314159], [5.0, 0.314159], seed=1)
        return cfg, veh, S.cold_start_inputs(cfg, veh, tr, x[:count], u[:count], 0.025), None, None
    if family == "lrn":
        tr = wl.synthetic_track("barc")
        cfg, veh = P.barc_lmpc(N, 5), P.barc_vehicle()
        laps = wl.synthetic_laps(tr, 5)
        x, u = wl.sample_states_near_laps(laps, BATCH, tr["L"], seed=0)
        inp = S.cold_start_inputs(cfg, veh, tr, x[:count], u[:count], 0.025)
        s_last, s0, L = inp["X_ref"][0, -1], inp["x_ic"][0], tr["L"]
        kk = np.abs(s0 - s_last) + L / 2      # lmpc_utils/utils.hpp:35-41, as bench.py forms the query
        q = np.stack([s_last + (kk - np.fmod(kk, L)) * np.sign(s0 - s_last), inp["X_ref"][1, -1]])
        ss_x, ss_j, _ = cbind.ss_query_batch(laps, L, cfg.num_ss_pts, cfg.num_ss_pts_per_lap, q)
        return cfg, veh, inp, ss_x, ss_j
    if family == "spc":
        # SURVEY.md 8(d) config 3 as written, = bench.py --workload lmpc --lmpc-data spec on rank 0: the five laps the tracking loop
        # records at speed scales 0.80 .. 1.0 (closed_loop.record_laps on the GPU; committed as DATA, tests/golden/spec_laps.npz, by
        # tests/golden/make_spec_laps.py -- tests/test_gpu_spec_workload.py holds a fresh recording to the file) and configs[1]'s x0
        tr = wl.synthetic_track("barc")
        cfg, veh = P.barc_lmpc(N, 5), P.barc_vehicle()
        laps = spec_laps()
        x, u = wl.sample_initial_states("barc", BATCH, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], seed=0)
        inp = S.cold_start_inputs(cfg, veh, tr, x[:count], u[:count], 0.025)
        q = ss_query_point(inp, tr["L"])
        ss_x, ss_j, _ = cbind.ss_query_batch(laps, tr["L"], cfg.num_ss_pts, cfg.num_ss_pts_per_lap, q)
        return cfg, veh, inp, ss_x, ss_j
    if family == "rec":
        import lmpc_scenario as LS

        veh, cfg, tr, laps, inp, q = LS.make(count, 21, N=N)
        ss_x, ss_j, _ = LS.oracle_safe_set(cfg, laps, q)
        return cfg, veh, inp, ss_x, ss_j
    raise KeyError(name)


def digest(inp: dict, ss_x=None, ss_j=None) -> np.ndarray:
    """One number per problem: the sum of |.| over every input array of the problem (drift of the regenerated inputs against the
    ones the fixture was solved on shows up here at the 1e-12 level, long before it shows in the comparison)."""
    d = sum(np.abs(np.asarray(inp[k], dtype=np.float64)).reshape(-1, np.asarray(inp[k]).shape[-1]).sum(axis=0) for k in INPUT_KEYS)
    if ss_x is not None:
        d = d + np.abs(ss_x).sum(axis=(0, 1)) + np.abs(ss_j).sum(axis=0)
    return d
