"""Round 4: the round-3 failure under test.  With the fp64 solve INLINED under the persistent loop of lmpc_cleanup_kernel the
<double, 7, 3> instance computed garbage, nondeterministically; behind a call it is bit for bit the
direct kernel.  This script hands whole batches of the N = 40 learning problem (KQ = 7; 160 points -> KS = 3, 96 -> KS = 2) to the
second pass (LMPC_DEBUG_CLEANUP_ALL=1, debug-hook builds) and compares with the direct fp64 kernel of the same build, several
times over.  One process per build variant (LMPC_HIP_LIBRARY), see scratch/r4_cleanup_rootcause.sh for the variants."""
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
dev = torch.device("cuda:0")
LIB = os.path.basename(os.environ.get("LMPC_HIP_LIBRARY", "liblmpc_hip.so"))


def setup(N, B, n_laps):
    tr = pkg.workloads.synthetic_track("barc")
    cfg = dict(pkg.presets.barc_lmpc(N, n_laps))
    laps = pkg.workloads.synthetic_laps(tr, n_laps)
    x, u = pkg.workloads.sample_states_near_laps(laps, B, tr["L"], seed=0)
    sv = pkg.Solver(cfg, pkg.presets.barc_vehicle(), device=0)
    sv.reserve(B)
    inp = sv.prepare(tr, x.T.copy(), 0.025)
    inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=dev)
    sv.set_safe_set(laps, tr["L"])
    s_last, s0, L = inp["X_ref"][0, -1], inp["x_ic"][0], tr["L"]
    kk = (s0 - s_last).abs() + L / 2
    q = torch.stack([s_last + (kk - torch.fmod(kk, L)) * torch.sign(s0 - s_last), inp["X_ref"][1, -1]]).contiguous()
    ss_x, ss_j, _ = sv.ss_query(q)

    def solve(mixed):
        out = sv.alloc_outputs(B)
        out["convex_combi_optm"] = torch.zeros((int(cfg["num_ss_pts"]), B), dtype=torch.float64, device=dev)
        o = sv.solve(inp, out, mixed=mixed, ss_x=ss_x, ss_j=ss_j)
        torch.cuda.synchronize()
        return {k: v.cpu().numpy().copy() for k, v in o.items() if hasattr(v, "cpu")}
    return solve


for N, B, nl in ((40, 2048, 5), (40, 2048, 3)):
    solve = setup(N, B, nl)
    a = solve(False)
    for rep in range(4):
        b = solve(True)
        diff = [k for k in ("X_optm", "U_optm", "dU_optm", "status", "iters") if not np.array_equal(a[k], b[k])]
        nbad = int((np.abs(a["X_optm"] - b["X_optm"]).max(axis=(0, 1)) > 0).sum())
        print(json.dumps({"lib": LIB, "N": N, "laps": nl, "rep": rep, "same_bits": not diff, "problems_differing": nbad,
                          "status_direct": np.bincount(a["status"], minlength=4).tolist(), "status_second_pass": np.bincount(b["status"], minlength=4).tolist(),
                          "max_dX": float(np.nanmax(np.abs(a["X_optm"] - b["X_optm"])))}), flush=True)
