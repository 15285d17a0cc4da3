"""Round 4: what a step costs when calls follow each other without a host sync (bench.py's one-stream loop) against a call
followed by a sync, per library build (LMPC_HIP_LIBRARY).  Large scratch frames turned out to cost launch time, not kernel time."""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
pkg.capi._ABI_SYMBOLS = tuple(s for s in pkg.capi._ABI_SYMBOLS if s != "lmpc_query_launch_for")
dev = torch.device("cuda:0")
LIB = os.path.basename(os.environ.get("LMPC_HIP_LIBRARY", "liblmpc_hip.so"))


def setup(kind, N, B):
    tr = pkg.workloads.synthetic_track("putnam" if kind == "iac" else "barc")
    kw = {}
    if kind == "lmpc":
        cfg, veh = dict(pkg.presets.barc_lmpc(N, 5)), pkg.presets.barc_vehicle()
        laps = pkg.workloads.synthetic_laps(tr, 5)
        x, u = pkg.workloads.sample_states_near_laps(laps, B, tr["L"], seed=0)
    elif kind == "iac":
        cfg, veh = dict(pkg.presets.iac_tracking_mpc(N)), pkg.presets.iac_vehicle()
        x, u = pkg.workloads.sample_initial_states("putnam", B, tr["L"], [-10.0, -0.314159], [5.0, 0.314159], seed=1)
    else:
        cfg, veh = dict(pkg.presets.barc_tracking_mpc(N)), pkg.presets.barc_vehicle()
        x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], seed=0)
    sv = pkg.Solver(cfg, veh, device=0)
    sv.reserve(B)
    inp = sv.prepare(tr, x.T.copy(), 0.025)
    inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=dev)
    if kind == "lmpc":
        sv.set_safe_set(laps, tr["L"])
        s_last, s0, L = inp["X_ref"][0, -1], inp["x_ic"][0], tr["L"]
        kk = (s0 - s_last).abs() + L / 2
        q = torch.stack([s_last + (kk - torch.fmod(kk, L)) * torch.sign(s0 - s_last), inp["X_ref"][1, -1]]).contiguous()
        ss_x, ss_j, _ = sv.ss_query(q)
        kw = dict(ss_x=ss_x, ss_j=ss_j)
    out = sv.alloc_outputs(B)
    if kind == "lmpc":
        out["convex_combi_optm"] = torch.zeros((int(cfg["num_ss_pts"]), B), dtype=torch.float64, device=dev)
    return sv, inp, out, kw


for kind, N, B, mixed in (("barc", 20, 4096, False), ("barc", 40, 4096, False), ("barc", 60, 4096, False), ("barc", 80, 4096, False),
                          ("lmpc", 20, 4096, False), ("lmpc", 20, 32768, True), ("iac", 40, 8192, True), ("lmpc", 40, 4096, False)):
    sv, inp, out, kw = setup(kind, N, B)
    fn = lambda: sv.solve(inp, out, mixed=mixed, **kw)  # noqa: E731
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    lat = []
    for _ in range(10):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    b2b = (time.perf_counter() - t0) / 20
    sv.enable_timing(True)
    fn()
    torch.cuda.synchronize()
    lin, qp = sv.last_kernel_ms()
    print(json.dumps({"lib": LIB, "case": "%s%d%s" % (kind, N, "m" if mixed else ""), "B": B, "call_plus_sync_ms": round(float(np.median(lat)) * 1e3, 3),
                      "back_to_back_ms": round(b2b * 1e3, 3), "kernels_ms": round(lin + qp, 3)}), flush=True)
    sv.close()
