"""Round-5 CPU probes: the bench's batches rebuilt with the oracle's restatement of the input preparation."""
import os, sys
os.environ.setdefault("OMP_NUM_THREADS", "1"); os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np, importlib
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from oracle import cbind, params as P, qp as Q, scenario as S
pkg = importlib.import_module("racing-lmpc-ros2_amd")

def batch(what, B=4096):
    """what: trkNN | iacNN | lrnNN -> cfg, veh, inp, ss_x, ss_j"""
    N = int(what[3:])
    if what.startswith("trk"):
        tr = pkg.workloads.synthetic_track("barc")
        cfg, veh = P.barc_tracking_mpc(N), P.barc_vehicle()
        x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], seed=0)
        return cfg, veh, S.cold_start_inputs(cfg, veh, tr, x, u, 0.025), None, None
    if what.startswith("iac"):
        tr = pkg.workloads.synthetic_track("putnam")
        cfg, veh = P.iac_tracking_mpc(N), P.iac_vehicle()
        x, u = pkg.workloads.sample_initial_states("putnam", B, tr["L"], [-10.0, -0.314159], [5.0, 0.314159], seed=1)
        return cfg, veh, S.cold_start_inputs(cfg, veh, tr, x, u, 0.025), None, None
    tr = pkg.workloads.synthetic_track("barc")
    cfg, veh = P.barc_lmpc(N, 5), P.barc_vehicle()
    laps = pkg.workloads.synthetic_laps(tr, 5)
    x, u = pkg.workloads.sample_states_near_laps(laps, B, tr["L"], seed=0)
    inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
    s_last, s0, L = inp["X_ref"][0, -1], inp["x_ic"][0], tr["L"]
    kk = np.abs(s0 - s_last) + L / 2
    q = np.stack([s_last + (kk - np.fmod(kk, L)) * np.sign(s0 - s_last), inp["X_ref"][1, -1]])
    ss_x, ss_j, nf = cbind.ss_query_batch(laps, L, cfg.num_ss_pts, cfg.num_ss_pts_per_lap, q)
    return cfg, veh, inp, ss_x, ss_j
