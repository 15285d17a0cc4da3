"""The C++ RacingMPC facade (reference class surface) end to end on the GPU."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
LIB = ROOT / "racing-lmpc-ros2_amd" / "lib"

pytestmark = pytest.mark.gpu


def _dm(f, a):
    a = np.atleast_2d(np.asarray(a, dtype=np.float64))
    f.write(f"{a.shape[0]} {a.shape[1]}\n")
    f.write(" ".join(repr(float(v)) for v in a.T.reshape(-1)) + "\n")  # column-major


def test_facade_solves_like_the_reference_class(golden, tmp_path):
    g = golden("qp_barc_tracking_n20")
    exe = LIB / "test_facade"
    assert exe.exists(), "run __graft_entry__.build() first"
    for b in (0, 7):
        p = tmp_path / f"problem{b}.txt"
        with open(p, "w") as f:
            f.write("20\n")
            _dm(f, g["x_ic"][:, b:b + 1])
            _dm(f, g["u_ic"][:, b:b + 1])
            _dm(f, g["X_ref"][:, :, b])
            _dm(f, g["U_ref"][:, :, b])
            for k in ("T_ref", "bound_left", "bound_right", "curvatures", "vel_ref"):
                _dm(f, g[k][:, b][None, :])
            _dm(f, g["X_optm"][:, :, b])
            _dm(f, g["U_optm"][:, :, b])
        r = subprocess.run([str(exe), str(p)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "PASS" in r.stdout, (r.stdout, r.stderr)


def test_facade_lmpc_loads_laps_queries_and_records(golden, tmp_path):
    """config.learning: SafeSetRecorder::load of the reference's lap files, SafeSetManager::query on the device,
    the LMPC solve against the dense optimum, and the recorder's lap segmentation + file output."""
    g = golden("qp_barc_lmpc_n20")
    exe = LIB / "test_facade_lmpc"
    assert exe.exists(), "run __graft_entry__.build() first"
    b = 3
    p = tmp_path / "lmpc_problem.txt"
    nf = 96  # three laps x 32 points: no padding in this scenario
    with open(p, "w") as f:
        f.write("20 %r\n" % float(g["L"]))
        _dm(f, g["x_ic"][:, b:b + 1])
        _dm(f, g["u_ic"][:, b:b + 1])
        _dm(f, g["X_ref"][:, :, b])
        _dm(f, g["U_ref"][:, :, b])
        for k in ("T_ref", "bound_left", "bound_right", "curvatures", "vel_ref"):
            _dm(f, g[k][:, b][None, :])
        _dm(f, g["X_optm"][:, :, b])
        _dm(f, g["U_optm"][:, :, b])
        _dm(f, g["ss_x"][:, :nf, b])
        _dm(f, g["ss_j"][:nf, b][None, :])
    laps = [str(ROOT / "tests" / "golden" / "barc_ss" / f"ss_lap_{i}") for i in (1, 2, 3)]
    r = subprocess.run([str(exe), str(p), *laps, str(tmp_path) + "/rec_"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("PASS"), (r.stdout[-2000:], r.stderr[-2000:])


def test_c_abi_from_plain_cpp_without_python_buffers():
    """bench_cabi: the reference's track file -> C++ RacingTrajectory -> device tables -> lmpc_prepare_batch ->
    lmpc_solve_batch, all from a C++ program holding its own HIP buffers (no torch anywhere in that process)."""
    exe = LIB / "bench_cabi"
    assert exe.exists(), "run __graft_entry__.build() first"
    track = ROOT / "tests" / "golden" / "barc_track" / "15_barc_optm.txt"
    r = subprocess.run([str(exe), str(track), "4096", "20"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    f = r.stdout.split()
    rate, solved = float(f[f.index("solves/s") - 1]), float(f[f.index("solved") + 1])
    assert solved > 0.99 and rate > 1e6, r.stdout
