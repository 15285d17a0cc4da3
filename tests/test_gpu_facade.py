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


@pytest.mark.parametrize("N,mode", [(20, "continuous"), (20, "step"), (60, "continuous")])
def test_node_core_drives_two_laps_of_the_reference_barc_track(N, mode):
    """RacingMPCNodeCore::step = RacingMPCNode::on_step_timer without ROS 2 (racing_mpc_node.cpp:150-477): global pose in,
    actuation out, first solve by the full-dynamics controller, first QP solve discarded (jit), plan shifted every period.
    Closed loop with a host plant on 15_barc_optm.txt at velocity_profile_scale 0.9, both step modes, and at the shipped
    horizon N = 60 (barc_tracking_mpc.param.yaml)."""
    import re
    exe = LIB / "test_node_core"
    assert exe.exists(), "run __graft_entry__.build() first"
    track = ROOT / "tests" / "golden" / "barc_track" / "15_barc_optm.txt"
    r = subprocess.run([str(exe), str(track), str(N), "2.1", mode], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PASS" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    m = re.search(r"laps ([\d.]+) time ([\d.]+) published (\d+) failed (\d+) .* worst_excess (-?[\d.]+) mean_step_ms ([\d.]+)", r.stdout)
    laps, t, published, failed, excess, step_ms = float(m[1]), float(m[2]), int(m[3]), int(m[4]), float(m[5]), float(m[6])
    assert laps >= 2.1 and failed <= published // 100 and excess < 0.02
    assert step_ms < 25.0                      # one controller inside its 25 ms period, host staging included
    print(f"N={N} {mode}: {laps:.2f} laps in {t:.2f} s, {published} steps, {failed} failed, {step_ms:.2f} ms per step")


def test_host_model_matches_the_device_model(pkg):
    """single_track_model.cpp (what the node core steps on the host) against the device kernels: the cold-start rollout of
    lmpc_prepare_batch is the same sequence of discrete_dynamics calls."""
    import ctypes as C
    import torch
    so = C.CDLL(str(LIB / "liblmpc_racing_mpc.so"))
    fn = so.lmpc_host_discrete_dynamics
    capi = __import__("importlib").import_module(pkg.__name__ + ".capi")
    veh = capi._fill(capi.CVehicle(), dict(pkg.presets.barc_vehicle()))
    solver = pkg.Solver(pkg.presets.barc_tracking_mpc(12), pkg.presets.barc_vehicle(), device=0)
    tr = pkg.workloads.synthetic_track("barc")
    rng = np.random.default_rng(4)
    x0 = np.array([rng.uniform(0, tr["L"], 8), rng.uniform(-0.2, 0.2, 8), rng.normal(0, 0.05, 8), rng.uniform(0.6, 3.0, 8),
                   rng.normal(0, 0.05, 8), rng.normal(0, 0.3, 8)])
    inp = solver.prepare(tr, x0, 0.025)
    X, K = inp["X_ref"].cpu().numpy(), inp["curvatures"].cpu().numpy()
    u = np.array([1e-9, 1e-9])
    for b in range(8):
        for i in range(11):
            xn = np.zeros(6)
            fn(C.byref(veh), X[:, i, b].copy().ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p), C.c_double(K[i, b]),
               C.c_double(0.025), xn.ctypes.data_as(C.c_void_p))
            assert np.abs(xn - X[:, i + 1, b]).max() <= 1e-12 * max(1.0, np.abs(xn).max()), (b, i)
