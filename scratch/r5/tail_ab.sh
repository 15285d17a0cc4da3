#!/bin/bash
# the loop tail's workgroup shape (LMPC_LOOP_WAVES = 4 / 8 / 16 waves sharing a car's knots), A/B on the closed loop
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
for v in "" _w16 _w4; do
  echo "== liblmpc_hip$v.so"
  LMPC_HIP_LIBRARY=$PWD/racing-lmpc-ros2_amd/lib/liblmpc_hip$v.so python - <<'PY' 2>&1 | grep -v amdgpu
import sys, numpy as np, torch, time
from pathlib import Path
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
tab = pkg.workloads.track_from_file(Path("tests/golden/barc_track/15_barc_optm.txt"), 1024)
def states(B):
    rng = np.random.default_rng(3); s0 = rng.uniform(0, tab["L"], B)
    return np.stack([s0, rng.uniform(-0.08, 0.08, B), rng.normal(0, 0.03, B), rng.uniform(0.6, 0.95, B) * np.interp(s0, np.arange(1024) * tab["L"] / 1024, tab["vel"]), np.zeros(B), np.zeros(B)])
for N, B, steps, warm in ((20, 4096, 666, False), (20, 4096, 666, True), (20, 16384, 200, True), (60, 4096, 200, True)):
    sv = pkg.Solver(dict(pkg.presets.barc_tracking_mpc(N)), pkg.presets.barc_vehicle(), 0)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        r = pkg.closed_loop.run(sv, tab, torch.as_tensor(states(B), device="cuda"), torch.zeros((2, B), dtype=torch.float64, device="cuda"), steps=steps, speed_scale=0.9, graph=True, warm=warm)
        torch.cuda.synchronize(); best = min(best, time.time() - t0)
    sv.close()
    print("N = %d, %d cars, %s: %.2f M car-steps/s (%.3f ms per period)" % (N, B, "warm" if warm else "cold", B * steps / best / 1e6, best / steps * 1e3), flush=True)
PY
done
