"""(GPU) Record the five laps SURVEY.md 8(d) config 3 stores in the safe set -- "running config 1's tracking loop for 5 laps with
seed-indexed speed scales {0.80, 0.85, 0.90, 0.95, 1.0}" -- exactly as bench.py --workload lmpc --lmpc-data spec does
(closed_loop.record_laps on a BARC tracking handle, N = 20), and store them as DATA: tests/golden/spec_laps.npz (lap0 .. lap4,
[n][6] each, oldest = slowest first; L).  The dense fixtures of the `spc` family (tests/dense_cases.py) and the CPU tests are
built on this file; tests/test_gpu_spec_workload.py re-records the laps on the GPU box and holds them to the file.

Run on a GPU box from the repo root:  python tests/golden/make_spec_laps.py [out.npz]
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

if __name__ == "__main__":
    pkg = load_package()
    out = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "tests" / "golden" / "spec_laps.npz"
    tr = pkg.workloads.synthetic_track("barc")
    sv = pkg.Solver(pkg.presets.barc_tracking_mpc(20), pkg.presets.barc_vehicle(), device=0)
    laps = pkg.closed_loop.record_laps(sv, tr)
    sv.close()
    out.parent.mkdir(parents=True, exist_ok=True)
    np.savez(out, L=np.float64(tr["L"]), **{"lap%d" % i: lap for i, lap in enumerate(laps)})
    print("spec laps:", [lap.shape for lap in laps], "->", out)
