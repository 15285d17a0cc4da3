"""Helper of tests/test_gpu_mixed_lmpc.py (run as a subprocess with LMPC_HIP_LIBRARY = the debug build of the library and
LMPC_DEBUG_CLEANUP_ALL=1 in the environment, which that build reads once per process): the fp64 second pass of lmpc_solve_batch_mixed handed the WHOLE batch, against the direct
fp64 kernel of lmpc_solve_batch -- same problems, so the same bits.  Prints one JSON line per case."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
dev = torch.device("cuda:0")


def learning(N, B, n_laps, mixed):
    tr = pkg.workloads.synthetic_track("barc")
    cfg = dict(pkg.presets.barc_lmpc(N, n_laps))
    laps = pkg.workloads.synthetic_laps(tr, n_laps)
    x, u = pkg.workloads.sample_states_near_laps(laps, B, tr["L"], seed=0)
    sv = pkg.Solver(cfg, pkg.presets.barc_vehicle(), device=0)
    inp = sv.prepare(tr, x.T.copy(), 0.025)
    inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=dev)
    sv.set_safe_set(laps, tr["L"])
    s_last, s0, L = inp["X_ref"][0, -1], inp["x_ic"][0], tr["L"]
    kk = (s0 - s_last).abs() + L / 2
    q = torch.stack([s_last + (kk - torch.fmod(kk, L)) * torch.sign(s0 - s_last), inp["X_ref"][1, -1]]).contiguous()
    ss_x, ss_j, _ = sv.ss_query(q)
    out = sv.alloc_outputs(B)
    out["convex_combi_optm"] = torch.zeros((int(cfg["num_ss_pts"]), B), dtype=torch.float64, device=dev)
    o = sv.solve(inp, out, mixed=mixed, ss_x=ss_x, ss_j=ss_j)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy().copy() for k, v in o.items() if hasattr(v, "cpu")}


def tracking(N, B, mixed):
    tr = pkg.workloads.synthetic_track("barc")
    x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], seed=0)
    sv = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), device=0)
    sv.set_waves_per_problem(1)  # the second pass IS the one-wave kernel; from N = 41 on a direct solve takes the two-wave kernel by itself (round 6)
    inp = sv.prepare(tr, x.T.copy(), 0.025)
    inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=dev)
    o = sv.solve(inp, sv.alloc_outputs(B), mixed=mixed)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy().copy() for k, v in o.items() if hasattr(v, "cpu")}


B = 1024
for N, n_laps in ((20, 3), (20, 5), (20, 0), (40, 0), (60, 0), (80, 0)):   # the (KQ, KS) the second pass is built for
    a, b = (learning(N, B, n_laps, False), learning(N, B, n_laps, True)) if n_laps else (tracking(N, B, False), tracking(N, B, True))
    same = all(np.array_equal(a[k], b[k]) for k in ("X_optm", "U_optm", "dU_optm", "status", "iters"))
    print(json.dumps({"N": N, "laps": n_laps, "same_bits": bool(same), "solved": int((a["status"] == 0).sum()),
                      "solved_second_pass": int((b["status"] == 0).sum())}), flush=True)
