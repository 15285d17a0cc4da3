import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
pkg = load_package()
for N in (10, 16, 18, 19, 20, 24, 40, 60, 80):
    s = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), 0)
    print(N, s.launch_info())
