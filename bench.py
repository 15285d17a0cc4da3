"""bench.py -- QP solves/sec of the batched tracking-MPC solve path on MI355X.

A step is one pass of the hot path (lmpc_solve_batch: stage linearisation + QP solve) over one
batch of synthetic problems already resident in HBM.  Workload = BASELINE.json configs[1]:
BARC tracking MPC, batch 4096 random x0, N = 20, fp64, per GPU (weak scaling across ranks).
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

# RCCL / CUDA-tensor sharing between the ranks of one node needs dmabuf IPC on this driver (the launcher exports it
# already; kept here so a bare `python -m torch.distributed.run ... bench.py` works too)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

ALGO_BYTES_PER_SOLVE = (13 * 20 + 5) * 8 + (10 * 20 - 4) * 8 + 8  # 3696 B at N = 20 fp64
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def measured_traffic_bytes(pmc_file="r06_pmc_tracking.json", kernel="lmpc_solve_kernel<double, 4, 0"):
    """HBM bytes per launch of the QP kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/r06_pmc_tracking.json: FETCH_SIZE and WRITE_SIZE are reported in KiB, collected in separate --pmc runs).
    The gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md applies to wide coalesced streams only; this
    kernel's reads are 8-byte strided or L2/MALL-resident workspace lines, so the raw counter is reported."""
    try:
        pmc = json.load(open(ROOT / "profiles" / pmc_file))
        k = next(v for n, v in pmc.items() if n.startswith(kernel))
        return (k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
    except Exception:
        return None


PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",),
              ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU"),
              ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"))


def live_counters(workload_argv, kernel, passes=PMC_PASSES):
    """Hardware counters of the QP kernel measured NOW: one rocprofv3 --pmc pass of this same script per counter group (FETCH_SIZE
    and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md; counters only, no trace domains), --steps 5, one stream; the average
    over the kernel's dispatches per counter.  {} when rocprofv3 is missing, this process is itself being profiled, or a pass
    fails (a failed group is left out, the others are kept)."""
    import shutil
    import signal
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}
    # this process is itself running under a profiler (the driver, or scratch/prof.sh): do not nest a second one
    if any(k in os.environ for k in ("ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_LIBRARY_CTOR", "ROCPROF_OUTPUT_PATH")) or \
            "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return {}
    got = {}
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["TMPDIR"] = "/tmp"
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            for n, group in enumerate(passes):
                out = os.path.join(d, "pass%d" % n)
                cmd = [exe, "--pmc", *group, "-d", out, "-o", "run", "--", sys.executable, str(ROOT / "bench.py"), "--steps", "5",
                       "--warmup", "1", "--no-cpu-baseline", "--no-batch1", "--no-others", "--no-latency", "--streams", "1", "--no-pmc", "--min-window", "0"] + workload_argv
                proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
                try:
                    proc.wait(timeout=90)
                except subprocess.TimeoutExpired:
                    os.killpg(proc.pid, signal.SIGKILL)  # the exact process group started above
                    continue
                dbs = [os.path.join(r, f) for r, _, fs in os.walk(out) for f in fs if f.endswith("_results.db")]
                if proc.returncode != 0 or not dbs:
                    continue
                con = sqlite3.connect(dbs[0])
                for counter in group:
                    row = con.execute("select avg(value), count(*) from counters_collection where counter_name = ? and kernel_name like ?",
                                      (counter, "%" + kernel + "%")).fetchone()
                    if row and row[1]:
                        got[counter] = float(row[0])
                con.close()
    except Exception:
        pass
    return got


def live_traffic_bytes(counters):
    """HBM bytes per launch from FETCH_SIZE + WRITE_SIZE (KiB each; units and corrections as measured_traffic_bytes)."""
    if "FETCH_SIZE" not in counters or "WRITE_SIZE" not in counters:
        return None
    return (counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024.0


ENGINE_CLOCK_GHZ = 2.4  # MI355X peak engine clock (MI355X_MICROARCH.md); the busy fractions below are against THAT clock


def live_engines(counters, kernel_ms):
    """What actually bounds the QP kernel, from the counter passes of THIS run (until round 5 the line carried
    constants read from a committed profile): busy fractions of the VALU (SQ_ACTIVE_INST_VALU is in 4-cycle units summed over
    waves; one VALU per SIMD, 4 SIMDs x 256 CUs) and of the LDS pipeline (SQ_LDS_IDX_ACTIVE: cycles summed over the 256 CU-local
    pipelines) over the kernel's duration as measured in this run, instructions per solve (= per wave), and the share of LDS cycles
    lost to bank conflicts."""
    need = ("SQ_ACTIVE_INST_VALU", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVES")
    if not all(k in counters for k in need) or not counters["SQ_WAVES"]:
        return None
    cyc = kernel_ms * 1e-3 * ENGINE_CLOCK_GHZ * 1e9
    e = {"valu_busy_frac": counters["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * 256 * 4), "lds_busy_frac": counters["SQ_LDS_IDX_ACTIVE"] / (cyc * 256),
         "valu_insts_per_solve": counters["SQ_INSTS_VALU"] / counters["SQ_WAVES"], "lds_insts_per_solve": counters["SQ_INSTS_LDS"] / counters["SQ_WAVES"],
         "source": "rocprofv3 --pmc passes of this command in this run, against the kernel's %.3f ms at %.1f GHz" % (kernel_ms, ENGINE_CLOCK_GHZ)}
    if "SQ_LDS_BANK_CONFLICT" in counters and counters["SQ_LDS_IDX_ACTIVE"]:
        e["lds_bank_conflict_frac"] = counters["SQ_LDS_BANK_CONFLICT"] / counters["SQ_LDS_IDX_ACTIVE"]
    return e


def engine_utilisation(kernel_ms, pmc_file="r06_pmc_tracking.json", kernel="lmpc_solve_kernel<double, 4, 0"):
    """What actually bounds the QP kernel: busy fractions of the FP64 VALU and of the LDS pipeline from the same
    committed PMC passes (SQ_ACTIVE_INST_VALU is in 4-cycle units summed over waves, one VALU per SIMD, 4 SIMDs x
    256 CUs; SQ_LDS_IDX_ACTIVE in cycles summed over the 256 CU-local LDS pipelines), against the kernel duration of
    that profile run.  Reported next to the HBM figure because the path is not HBM-bound."""
    try:
        pmc = json.load(open(ROOT / "profiles" / pmc_file))
        k = next(v for n, v in pmc.items() if n.startswith(kernel))
        prof_ms = pmc.get("_meta", {}).get("qp_kernel_ms", kernel_ms)
        cyc = prof_ms * 1e-3 * pmc.get("_meta", {}).get("clock_ghz", 2.3) * 1e9
        return {"valu_busy_frac": k["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * 256 * 4),
                "lds_busy_frac": k["SQ_LDS_IDX_ACTIVE"] / (cyc * 256),
                "valu_insts_per_solve": k["SQ_INSTS_VALU"] / k["SQ_WAVES"], "lds_insts_per_solve": k["SQ_INSTS_LDS"] / k["SQ_WAVES"],
                "source": "profiles/%s (rocprofv3 --pmc, batch 4096, N 20)" % pmc_file}
    except Exception:
        return None


def shard_bounds(total: int, world: int, rank: int):
    """Contiguous slice [lo, hi) of `total` problems owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def packed_numel(N: int, B: int) -> int:
    return (10 * N - 4 + 2) * B  # X 6N + U 2(N-1) + dU 2(N-1) per problem, + status and iters (results AND status)


def pack_results(out: dict, flat):
    """X_optm | U_optm | dU_optm | status | iters of this rank's slice into one contiguous buffer for the gather (the two
    integer arrays ride along in the buffer's floating type -- exact for these ranges -- so that one collective moves all of
    it and rank 0 can tell which problems of which rank failed)."""
    import torch

    dt = flat.dtype
    torch.cat([out["X_optm"].reshape(-1), out["U_optm"].reshape(-1), out["dU_optm"].reshape(-1), out["status"].to(dt), out["iters"].to(dt)],
              out=flat)
    return flat


def unpack_results(gbuf, world: int, N: int, B: int) -> dict:
    """Inverse of pack_results over the all-gathered buffer: arrays with the global batch axis last."""
    import torch

    per = gbuf.reshape(world, packed_numel(N, B))
    nX, nU = 6 * N * B, 2 * (N - 1) * B
    o = nX + 2 * nU
    X = torch.cat([per[r, :nX].reshape(6, N, B) for r in range(world)], dim=2)
    U = torch.cat([per[r, nX:nX + nU].reshape(2, N - 1, B) for r in range(world)], dim=2)
    dU = torch.cat([per[r, nX + nU:o].reshape(2, N - 1, B) for r in range(world)], dim=2)
    status = torch.cat([per[r, o:o + B] for r in range(world)]).to(torch.int32)
    iters = torch.cat([per[r, o + B:o + 2 * B] for r in range(world)]).to(torch.int32)
    return {"X_optm": X, "U_optm": U, "dU_optm": dU, "status": status, "iters": iters}


def usable_cores():
    """Host threads this process can actually keep busy: the affinity mask, capped by the cgroup CPU quota (a container
    that shows 256 CPUs may be allowed 16 CPUs' worth of time; oversubscribing the quota only adds throttling)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]      # cgroup v2
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())   # cgroup v1
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def cpu_baseline(pkg, N, batch, workload="tracking", laps=None, n_laps=5, budget_s=8.0):
    """Time the C restatement (oracle, 'port') on the host cores over a bounded sample of the SAME workload the GPU line is quoted
    on: tracking (configs[1]), iac (configs[3]'s problem), lmpc (configs[2] / configs[4]: the safe-set query + the learning QP on
    `laps`; the twin has no error-dynamics regression, so configs[4]'s CPU figure is the solve without it and says so)."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    from oracle import cbind, params as OP, qp as OQ, scenario as OS

    lmpc, iac = workload == "lmpc", workload == "iac"
    if iac:
        veh, cfg, kind, seed = OP.iac_vehicle(), OP.iac_tracking_mpc(N), "putnam", 1
    elif lmpc:
        veh, cfg, kind, seed = OP.barc_vehicle(), OP.barc_lmpc(N, n_laps), "barc", 0
    else:
        veh, cfg, kind, seed = OP.barc_vehicle(), OP.barc_tracking_mpc(N), "barc", 0
    tr = pkg.workloads.synthetic_track(kind)
    u_lo, u_hi, _, _ = OQ.effective_bounds(cfg, veh)
    cores = usable_cores()
    per_thread = 32
    B = cores * per_thread          # same distribution as the GPU batch, sized so every thread gets a slice
    x, u = pkg.workloads.sample_initial_states(kind, B, tr["L"], u_lo, u_hi, seed)
    inp = OS.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
    ss = [None, None]
    lam = None
    if lmpc:
        s_last, s0, L = inp["X_ref"][0, -1], inp["x_ic"][0], tr["L"]
        kk = np.abs(s0 - s_last) + L / 2
        q = np.stack([s_last + (kk - np.fmod(kk, L)) * np.sign(s0 - s_last), inp["X_ref"][1, -1]])
        ssx, ssj, _ = cbind.ss_query_batch([np.asarray(a, dtype=np.float64) for a in laps], L, cfg.num_ss_pts, cfg.num_ss_pts_per_lap, q)
        ss = [np.ascontiguousarray(ssx), np.ascontiguousarray(ssj)]
        lam = np.zeros((cfg.num_ss_pts, B))
    # straight into the C entry point with shared, preallocated arrays: every thread solves its own slice of the batch
    # and writes its own slice of the outputs (ctypes releases the GIL for the duration of the call)
    lib = cbind.lib()
    keys = ("x_ic", "u_ic", "X_ref", "U_ref", "T_ref", "bound_left", "bound_right", "curvatures", "vel_ref")
    arrs = [np.ascontiguousarray(inp[k], dtype=np.float64) for k in keys]
    X, U, dU = np.zeros((6, N, B)), np.zeros((2, N - 1, B)), np.zeros((2, N - 1, B))
    status, iters, kkt = np.full(B, -1, dtype=np.int32), np.zeros(B, dtype=np.int32), np.zeros((4, B))
    cc, cv = cbind.c_config(cfg), cbind.c_vehicle(veh)
    ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731

    def solve_range(b0, b1):
        rc = lib.lmpc_oracle_solve_range(C.byref(cc), C.byref(cv), C.c_int32(B), C.c_int32(b0), C.c_int32(b1),
                                         *[ptr(a) for a in arrs], ptr(ss[0]), ptr(ss[1]), ptr(X), ptr(U), ptr(dU), ptr(lam),
                                         ptr(status), ptr(iters), ptr(kkt))
        assert rc == 0, rc

    solve_range(0, per_thread)          # cold: page faults of the work area, instruction cache
    t0 = time.perf_counter()
    solve_range(0, per_thread)          # the figure quoted: a warm call
    per = (time.perf_counter() - t0) / per_thread
    def run(reps):
        def work(c):
            for _ in range(reps):
                solve_range(c * per_thread, (c + 1) * per_thread)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(work, range(cores)))
        return time.perf_counter() - t0

    cal = run(2)                                   # calibration: how the host really scales with all threads busy
    reps = int(min(4096, max(2, budget_s / (cal / 2))))  # then a sample of about `budget_s` seconds of wall clock
    dt = run(reps)
    assert (status == 0).mean() > 0.98, np.bincount(status, minlength=3)
    return {"value": reps * B / dt, "unit": "solves/s", "cores": cores, "kind": "port",
            "single_thread_solve_ms": per * 1e3,   # one thread, one problem at a time (the first 32 problems, second call, other cores idle)
            "sample": f"{reps} x {B} problems of the bench workload ({per_thread} per thread per call), static split "
                      f"over {cores} host threads = usable cores (affinity mask capped by the cgroup CPU quota; os.cpu_count() = {os.cpu_count()}) (one C call per slice, shared preallocated arrays), oracle/c/lmpc_oracle.c -O3 ({dt:.1f} s wall)"
                      + ("; the safe-set query is outside the timed loop, the twin has no error-dynamics regression (a configs[4] line's CPU figure is the learning solve alone)" if lmpc else "")}


def other_configs(steps, warmup):
    """The other BASELINE configs, each as a short leg of this same script in a child process (own handles, streams and
    buffers; --no-pmc, no CPU leg): configs[2] as quoted, and the per-GPU shares of the two 8-GPU configs.  Every entry
    carries what the headline carries: ms_per_step (pipelined), ms_per_step_one_stream, kernels_ms, roofline."""
    import subprocess
    legs = [("configs[2]", ["--workload", "lmpc", "--batch", "4096", "--horizon", "20"]),
            ("configs[3], share of one GPU (8192 of 65536)", ["--workload", "iac", "--horizon", "40", "--batch", "8192", "--precision", "f32"]),
            ("configs[4], share of one GPU (32768 of 262144)", ["--workload", "lmpc", "--batch", "32768", "--horizon", "20", "--precision", "mixed", "--regression"]),
            # configs[1]'s problem at the horizons the reference SHIPS for tracking (barc_tracking_mpc.param.yaml n = 60, iac_car_tracking_mpc.param.yaml
            # n = 80; BASELINE quotes N = 20): the two-wavefronts-per-problem kernels of round 6, so that their numbers are in the driver's record too
            ("configs[1]'s problem at the shipped BARC horizon (N = 60; not a BASELINE config)", ["--workload", "tracking", "--batch", "4096", "--horizon", "60"]),
            ("configs[1]'s problem at the shipped IAC horizon (N = 80; not a BASELINE config)", ["--workload", "tracking", "--batch", "4096", "--horizon", "80"]),
            # the friendlier learning workload of rounds 1 - 4 (analytic laps, states near the last lap), reported separately
            ("configs[2] on the rounds-1-4 workload (states near the laps; not SURVEY 8d's)", ["--workload", "lmpc", "--batch", "4096", "--horizon", "20", "--lmpc-data", "near"]),
            ("configs[4] share on the rounds-1-4 workload (states near the laps; not SURVEY 8d's)",
             ["--workload", "lmpc", "--batch", "32768", "--horizon", "20", "--precision", "mixed", "--regression", "--lmpc-data", "near"])]
    keep = ("metric", "value", "unit", "ms_per_step", "ms_per_step_one_stream", "value_one_stream", "timed_steps", "timed_window_s", "dtype", "config", "kernels_ms",
            "solved_fraction", "mean_ipm_iters", "p50_solve_ms", "p99_solve_ms", "ss_query_kernel", "cpu_baseline", "launch")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = []
    for name, argv in legs:
        # the three BASELINE legs carry what the headline carries: HBM traffic measured in the run (the FETCH_SIZE and
        # WRITE_SIZE passes only: --pmc-traffic-only) and the CPU twin timed on the same workload (a ~4 s sample); the two legs on the
        # rounds-1-4 workload are for continuity and carry neither
        full = name.startswith("configs[") and "rounds-1-4" not in name and "not a BASELINE config" not in name
        traffic_only = "not a BASELINE config" in name   # (the shipped-horizon legs: kernel time and HBM traffic, no CPU leg)
        cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", str(max(5, min(steps, 20))), "--warmup", str(max(1, min(warmup, 3))),
               "--no-batch1", "--no-others", "--min-window", "0.3"] + (["--pmc-traffic-only", "--cpu-budget", "4"] if full else (["--pmc-traffic-only", "--no-cpu-baseline"] if traffic_only else ["--no-cpu-baseline", "--no-pmc"])) + argv
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
            j = json.loads(line)
            e = {"baseline_config": name, **{k: j.get(k) for k in keep if k in j}}
            e["config"] = {k: v for k, v in (j.get("config") or {}).items() if k != "ranks_seen"}
            rf = j.get("roofline") or {}
            e["roofline"] = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "traffic_over_algorithmic", "algorithmic_bytes_per_solve")}
            out.append(e)
        except Exception as ex:  # a failed leg is reported, not hidden
            out.append({"baseline_config": name, "error": "%s: %s" % (type(ex).__name__, str(ex)[:300])})
    return out


def closed_loop_leg(pkg, dev, cars=4096, periods=400):
    """SURVEY.md 8(f) rank 1 measured next to the headline: `cars` cars in closed loop on the synthetic BARC-scale track, one
    control period = linearisation + QP + lmpc_loop_advance_batch (input selection, plant step, shift / cold restart, statistics),
    captured as a HIP graph; cold solves and the active-set warm start on the shifted previous plan."""
    import torch
    tr = pkg.workloads.synthetic_track("barc")
    rng = np.random.default_rng(3)
    s0 = rng.uniform(0, tr["L"], cars)
    vel = np.interp(s0, np.arange(len(tr["vel"])) * tr["L"] / len(tr["vel"]), np.asarray(tr["vel"]))
    x0 = torch.as_tensor(np.stack([s0, rng.uniform(-0.08, 0.08, cars), rng.normal(0, 0.03, cars), rng.uniform(0.6, 0.9, cars) * vel, np.zeros(cars), np.zeros(cars)]), device=dev)
    u0 = torch.zeros((2, cars), dtype=torch.float64, device=dev)
    out = {"cars": cars, "periods": periods, "horizon": 20, "unit": "car-steps/s",
           "period": "linearise + QP + lmpc_loop_advance_batch (plant RK4 x 2, shift or cold restart, statistics); HIP graph"}
    try:
        for warm in (False, True):
            sv = pkg.Solver(pkg.presets.barc_tracking_mpc(20), pkg.presets.barc_vehicle(), device=dev.index)
            best, r = None, None
            for _ in range(2):  # (the first run pays the capture and the workspace)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r = pkg.closed_loop.run(sv, tr, x0, u0, steps=periods, speed_scale=0.9, graph=True, warm=warm)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            sv.close()
            key = "warm" if warm else "cold"
            out[key] = {"value": cars * periods / best, "ms_per_period": best / periods * 1e3, "cars_with_a_failed_solve": int((r["n_fail"] > 0).sum()),
                        # the cars are started at 0.6 .. 0.9 of the speed profile wherever they are on the track; a few per cent of them are then too
                        # fast into a corner for the controller's HARD vx box (faithful to the reference, racing_mpc.cpp:147) and take the cold-restart
                        # branch once or twice in their first periods (profiles/r05_closed_loop_failures.txt: the dense oracle solves none of them)
                        "failed_solve_share": float(r["n_fail"].sum()) / (cars * periods), "cars_restarted_share": float((r["n_fail"] > 0).float().mean()),
                        "median_laps": float(r["distance"].median()) / float(tr["L"])}
            if warm:
                out[key]["warm_starts_accepted"] = r["warm_hit_rate"]
    except Exception as ex:  # reported, not hidden
        out["error"] = "%s: %s" % (type(ex).__name__, str(ex)[:300])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--workload", choices=["tracking", "lmpc", "iac"], default="tracking",
                    help="tracking = BASELINE configs[1] (the quoted metric); lmpc = configs[2] (5-lap safe set); "
                         "iac = configs[3]'s problem (IAC/Putnam tracking, use --horizon 40 --batch 8192) in fp64")
    ap.add_argument("--precision", choices=["f64", "f32", "mixed"], default="f64",
                    help="f32: lmpc_solve_batch_f32 (BASELINE configs[3] as quoted: --workload iac --horizon 40 --batch 8192 --precision f32); "
                         "mixed: lmpc_solve_batch_mixed, fp64 arrays around an fp32 interior-point iteration (BASELINE configs[4]: "
                         "--workload lmpc --batch 32768 --precision mixed)")
    ap.add_argument("--streams", type=int, default=3,
                    help="consecutive steps alternate between this many HIP streams (one handle, workspace and output buffer "
                         "each), so the tail of one batch overlaps the head of the next; 1 = strictly one batch at a time")
    ap.add_argument("--launch-order", choices=["default", "previous"], default="default",
                    help="previous: start the QP kernel's workgroups longest-first by the iteration counts of the previous solve of "
                         "the same batch (lmpc_set_launch_order) -- what a closed loop has; the bench repeats one batch, so here the "
                         "prediction is perfect, and the figure is labelled as such (never the default)")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL result gather for N > 1")
    ap.add_argument("--regression", action="store_true", help="--workload lmpc only: switch the error-dynamics regression on "
                    "(safe_set.cpp:182-245; BASELINE configs[4]: 'LMPC + error-dynamics residual term') -- 2200 recorded sample pairs "
                    "from a plant with 15 %% less grip, every stage of every problem regressed before its QP")
    ap.add_argument("--lmpc-data", choices=["spec", "near"], default="spec",
                    help="--workload lmpc: spec (default) = SURVEY.md 8d config 3 as written: the five laps are produced by the tracking "
                         "loop at speed scales 0.80 .. 1.0 (closed_loop.record_laps) and the 4096 queries are config 2's random x0; "
                         "near = the friendlier workload of rounds 1 - 4 (analytic laps, states drawn near the last lap), kept for continuity")
    ap.add_argument("--laps", type=int, choices=[3, 5], default=5,
                    help="--workload lmpc: laps stored in the safe set, 32 points each: 5 (default) = SURVEY.md 8d config 3 (160 points), "
                         "3 = the value the reference ships (barc_lmpc.param.yaml: max_lap_stored 3, num_ss_pts 96; the newest three of the five laps)")
    ap.add_argument("--laps-npz", default=None, help=argparse.SUPPRESS)  # (the counter passes nested in a run: the laps the parent recorded,
    #                                                                       so that the profiled child does not drive the tracking loop again)
    ap.add_argument("--ss-mode", choices=["idx", "arrays"], default="idx",
                    help="--workload lmpc: idx (default) = the safe set by reference (lmpc_ss_query_idx_batch + lmpc_solve_batch_ss_idx: 640 B of "
                         "codes per query); arrays = (ss_x, ss_j) materialised per query (8960 B), the interface of rounds 1 - 4")
    ap.add_argument("--min-window", type=float, default=0.5, help="the timed region repeats the --steps block until it spans at least this "
                    "many seconds AND at least 200 steps (VERDICT r4 item 9a: 20 steps of 0.7 ms are a 15 ms window); 0 = exactly --steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the two rocprofv3 --pmc passes that measure the QP kernel's HBM "
                    "traffic (N = 1 only, ~30 s); roofline.traffic then comes from the committed passes in profiles/")
    ap.add_argument("--pmc-traffic-only", action="store_true", help="of the counter passes, only FETCH_SIZE and WRITE_SIZE (the `others` legs of the default run)")
    ap.add_argument("--cpu-budget", type=float, default=8.0, help="seconds of wall clock the CPU twin is timed for")
    ap.add_argument("--no-batch1", action="store_true", help="skip the single-car latency probe (profiling runs: keeps every "
                                                             "launch of the QP kernel at the bench batch size)")
    ap.add_argument("--output-layout", choices=["soa", "aos"], default="soa", help="result arrays [component][knot][batch] (default, what the "
                    "batch-parallel consumers read) or [batch][knot][component] (lmpc_set_output_layout: full-line stores; never the headline)")
    ap.add_argument("--no-latency", action="store_true", help="skip the per-call latency loop (counter passes: every launch of the QP "
                    "kernel is then one of the timed steps; kernels_ms comes from 20 calls)")
    ap.add_argument("--no-others", action="store_true", help="default run only: skip the short legs on the other BASELINE configs "
                    "(configs[2], the per-GPU shares of configs[3] and configs[4]) that are reported under `others`")
    args = ap.parse_args()

    # --gpus is authoritative.  Under a launcher (WORLD_SIZE set) it must agree with the world the launcher made; started
    # bare with --gpus N > 1 the script re-executes itself under torch.distributed.run (one rank per GPU, rendezvous on
    # 127.0.0.1), so `python bench.py --gpus 8` cannot silently measure one GPU.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:])

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # pre-flight hook for a one-GPU box (tests/test_gpu_path.py): all ranks share device 0 and rendezvous over gloo,
    # so the N > 1 control flow (rank env, per-rank workload, barriers, max-over-ranks) runs without RCCL
    shared_gpu = os.environ.get("LMPC_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    if not shared_gpu:
        assert local < torch.cuda.device_count(), f"rank {rank}: LOCAL_RANK {local} but {torch.cuda.device_count()} visible GPUs"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    assert torch.cuda.current_device() == local
    # which physical device every rank sits on (one distinct GPU per rank unless the one-GPU preflight shares device 0)
    props = torch.cuda.get_device_properties(local)
    me = {"rank": rank, "device": local, "uuid": str(getattr(props, "uuid", "")), "name": props.name}
    ranks_seen = [me]
    if world > 1:
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, me)
        if not shared_gpu:
            assert len({(r["device"], r["uuid"]) for r in ranks_seen}) == world, f"ranks share a GPU: {ranks_seen}"

    pkg = load_package()
    f32 = args.precision == "f32"
    mixed = args.precision == "mixed"
    N, B = args.horizon, args.batch
    lmpc = args.workload == "lmpc"
    iac = args.workload == "iac"
    tr = pkg.workloads.synthetic_track("putnam" if iac else "barc")
    if lmpc:
        cfgd = pkg.presets.barc_lmpc(N, args.laps)  # 5 laps stored, 32 per lap -> 160 points (3: the shipped 96)
        if args.laps_npz:
            with np.load(args.laps_npz) as z:
                laps = [z["lap%d" % i] for i in range(len(z.files))]
        elif args.lmpc_data == "spec":
            # "produced by running config 1's tracking loop for 5 laps with seed-indexed speed scales {0.80 .. 1.0}", 0.03 s samples.
            # With several ranks, rank 0 records and the others receive (the safe set is replicated, ~50 KB)
            laps = None
            if rank == 0:
                trk_sv = pkg.Solver(pkg.presets.barc_tracking_mpc(20), pkg.presets.barc_vehicle(), device=local)
                laps = pkg.closed_loop.record_laps(trk_sv, tr)
                trk_sv.close()
            if world > 1:
                box = [laps]
                dist.broadcast_object_list(box, src=0)
                laps = box[0]
        else:
            laps = pkg.workloads.synthetic_laps(tr, 5)
        laps = laps[-args.laps:]  # (oldest first: the manager keeps the newest max_lap_stored)

        reg_laps = []
        if args.regression:
            # recorded data of a plant that differs from the model: states around the stored laps, their successors one
            # 30 ms period later from the PLANT step kernel of a second handle (15 % less grip) -- two-sample laps
            # (rank 0 produces them, like the laps; the others receive)
            if rank == 0:
                pv = dict(pkg.presets.barc_vehicle())
                pv["mu"] *= 0.85
                plant = pkg.Solver(cfgd, pv, device=local)
                reg_laps = pkg.workloads.regression_sample_pairs(
                    tr, laps, lambda xa, ua: plant.plant_step(tr, torch.as_tensor(xa.T.copy(), device=dev), torch.as_tensor(ua.T.copy(), device=dev),
                                                              0.03).cpu().numpy().T)
                plant.close()
            if world > 1:
                box = [reg_laps]
                dist.broadcast_object_list(box, src=0)
                reg_laps = box[0]

        def make_solver():
            sv = pkg.Solver(cfgd, pkg.presets.barc_vehicle(), device=local)
            sv.set_safe_set(laps, tr["L"])
            if reg_laps:
                sv.set_regression_laps(reg_laps, dist_max=0.6)
            return sv
        solver = make_solver()
        if args.lmpc_data == "spec":   # "batch=4096 queries as config 2": config 2's random x0 (vx inside this controller's box)
            x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], seed=rank)
        else:
            x, u = pkg.workloads.sample_states_near_laps(laps, B, tr["L"], seed=rank)
    elif iac:
        def make_solver():
            return pkg.Solver(pkg.presets.iac_tracking_mpc(N), pkg.presets.iac_vehicle(), device=local)
        solver = make_solver()
        x, u = pkg.workloads.sample_initial_states("putnam", B, tr["L"], [-10.0, -0.314159], [5.0, 0.314159], seed=rank + 1)
    else:
        def make_solver():
            return pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), device=local)
        solver = make_solver()
        P = solver.config
        u_lo = [max(P["u_min"][0], -0.015), max(P["u_min"][1], -0.314159)]
        u_hi = [min(P["u_max"][0], 0.015), min(P["u_max"][1], 0.314159)]
        x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], u_lo, u_hi, seed=rank)
    solver.reserve(B)
    if args.output_layout == "aos":
        assert world == 1 and not f32, "--output-layout aos: single GPU, fp64 arrays (the gather packs the default layout)"
        _mk = make_solver
        def make_solver():  # noqa: E306
            sv = _mk()
            sv.set_output_layout("aos")
            return sv
        solver.set_output_layout("aos")
    inp = solver.prepare(tr, x.T.copy(), 0.025)   # node cold start on the device (racing_mpc_node.cpp:210-292)
    inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=dev)
    S = max(1, args.streams)
    solvers = [solver] + [make_solver() for _ in range(S - 1)]
    for sv in solvers[1:]:
        sv.reserve(B)
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else [None]
    outs = [solver.alloc_outputs(B) for _ in range(max(S, 2))]
    query = None
    if lmpc:
        for o in outs:
            o["convex_combi_optm"] = torch.zeros((cfgd["num_ss_pts"], B), dtype=torch.float64, device=dev)
        # query = last knot of the abscissa-aligned reference (racing_mpc.cpp:219-223,249-254)
        ss_idx_mode = args.ss_mode == "idx"
        ss_query_fn = (lambda sv, q, out=None: sv.ss_query_idx(q, out=out)) if ss_idx_mode else (lambda sv, q, out=None: sv.ss_query(q, out=out))
        ss_bufs = [ss_query_fn(solver, inp["X_ref"][:2, -1].contiguous()) for _ in outs]   # one result buffer set per output slot
        s_last, s0, L = inp["X_ref"][0, -1], inp["x_ic"][0], tr["L"]
        kk = (s0 - s_last).abs() + L / 2
        query = torch.stack([s_last + (kk - torch.fmod(kk, L)) * torch.sign(s0 - s_last), inp["X_ref"][1, -1]]).contiguous()
    gather = world > 1 and not args.no_gather
    if gather:
        import torch.distributed as dist
        gdt = torch.float32 if f32 else torch.float64
        flat = [torch.empty(packed_numel(N, B), dtype=gdt, device=dev) for _ in range(2)]
        gbuf = [torch.empty(world * flat[0].numel(), dtype=gdt, device=dev) for _ in range(2)]

    if f32:
        assert not lmpc, "single precision is built for the tracking problem"
        inp32 = {k: (v.to(torch.float32).contiguous() if hasattr(v, "to") else v) for k, v in inp.items()}
        outs = [solver.solve_f32(inp32) for _ in range(max(S, 2))]
    if gather:  # one staging / receive buffer per stream
        flat = [torch.empty_like(flat[0]) for _ in range(S)]
        gbuf = [torch.empty_like(gbuf[0]) for _ in range(S)]
    torch.cuda.synchronize()

    orders = {}
    if args.launch_order == "previous":
        for sv in solvers:
            orders[id(sv)] = torch.arange(B, dtype=torch.int32, device=dev)
            sv.set_launch_order(orders[id(sv)])

    def solve_step(k, sv=None):
        o = solve_step_(k, sv)
        if orders:
            (sv or solver).launch_order_from_iters(o["iters"], orders[id(sv or solver)])
        return o

    def solve_step_(k, sv=None):
        sv = sv or solver
        o = outs[k % len(outs)]
        if f32:
            sv.solve_f32(inp32, o)
        elif lmpc:
            if ss_idx_mode:
                idx, _ = sv.ss_query_idx(query, out=ss_bufs[k % len(ss_bufs)])
                sv.solve(inp, o, ss_idx=idx, mixed=mixed)
            else:
                ss_x, ss_j, _ = sv.ss_query(query, out=ss_bufs[k % len(ss_bufs)])
                sv.solve(inp, o, ss_x=ss_x, ss_j=ss_j, mixed=mixed)
        else:
            sv.solve(inp, o, mixed=mixed)
        return o

    pending = [None] * S   # the gather in flight on each stream's buffers

    def step(k):
        j = k % S
        with (torch.cuda.stream(streams[j]) if streams[j] is not None else contextlib.nullcontext()):
            o = solve_step(j, solvers[j])
            if gather:  # a collective: every rank must take this path the same number of times
                if pending[j] is not None:
                    pending[j].wait()
                f = pack_results(o, flat[j])
                pending[j] = dist.all_gather_into_tensor(gbuf[j], f, async_op=True)

    def drain():
        for j in range(S):
            if pending[j] is not None:
                pending[j].wait()
                pending[j] = None
        torch.cuda.synchronize()

    # every stream's handle once, untimed: first-launch costs (code object load, workspace growth, RCCL channel setup)
    # belong to initialisation whatever --warmup is
    for k in range(S):
        step(k)
    drain()
    for k in range(args.warmup):
        step(k)
    drain()
    # How many times the --steps block is repeated inside the timed region: at least 200 steps and --min-window seconds, from a
    # short untimed probe of the pipeline's pace (rank 0 decides for everybody: the steps are collectives when gathering)
    repeats = 1
    if args.min_window > 0:
        tp = time.perf_counter()
        for k in range(max(S, 4)):
            step(k)
        drain()
        est = (time.perf_counter() - tp) / max(S, 4)
        repeats = max(1, -(-max(200, int(1.3 * args.min_window / max(est, 1e-6)) + 1) // args.steps))  # (the probe is slower than the steady pipeline)
        if world > 1:
            box = [repeats]
            dist.broadcast_object_list(box, src=0)
            repeats = int(box[0])
    timed_steps = args.steps * repeats
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(timed_steps):
        step(k)
    drain()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    one_stream_value = None
    if world == 1 and S > 1:  # the same steps strictly one after the other, for reference
        # (on the default stream, which has not launched these kernels yet: its first launch grows that queue's scratch
        #  arena -- a one-off of ~20 ms at the 1.5 KB frames of the N = 60 kernels -- and belongs to initialisation like the
        #  first launches on the other streams)
        solve_step(0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(timed_steps):
            solve_step(k)
        torch.cuda.synchronize()
        one_stream_value = B * timed_steps / (time.perf_counter() - t1)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if shared_gpu else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- per-call latency distribution and the dominant kernel's duration (rank 0, untimed) ----
    lat, lin_ms, sol_ms = [], [], []
    if rank == 0:
        solver.enable_timing(True)
        # p99 over >= 1000 timed calls (bounded to ~3 s of GPU time for the big batches)
        n_lat = 20 if args.no_latency else int(max(100, min(1000, 3.0 / max(elapsed / timed_steps, 1e-4))))
        for k in range(n_lat):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            solve_step(0)   # rank-local: no collective in this loop
            e1.record()
            e1.synchronize()
            lat.append(e0.elapsed_time(e1))
            a, b = solver.last_kernel_ms()
            lin_ms.append(a)
            sol_ms.append(b)
        solver.enable_timing(False)
        ss_ms = []
        if lmpc:  # the safe-set kernel alone (both kernels timed separately and together)
            for k in range(60):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ss_query_fn(solver, query, out=ss_bufs[0])
                e1.record()
                e1.synchronize()
                if k >= 10:
                    ss_ms.append(e0.elapsed_time(e1))
        # one car: the latency a single controller sees against its 25 ms period
        lat1 = []
        if not lmpc and not args.no_batch1:
            inp1 = {k: (v[..., :1].contiguous() if hasattr(v, "dim") and v.dim() >= 1 else v) for k, v in inp.items()}
            out1 = solver.solve_f32(inp1) if f32 else solver.alloc_outputs(1)
            for k in range(250):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                (solver.solve_f32 if f32 else solver.solve)(inp1, out1)
                e1.record()
                e1.synchronize()
                if k >= 50:
                    lat1.append(e0.elapsed_time(e1))
        st = outs[0]["status"].cpu().numpy()
        iters = outs[0]["iters"].cpu().numpy()
    if world > 1:
        dist.barrier()

    gathered = None
    if gather and rank == 0:  # what rank 0 holds after the last gather: every rank's results and statuses
        full = unpack_results(gbuf[(timed_steps - 1) % S], world, N, B)
        gathered = {"problems": int(full["status"].numel()), "solved_fraction": float((full["status"] == 0).float().mean()),
                    "mean_iters": float(full["iters"].float().mean())}
    if rank == 0:
        value = world * B * timed_steps / elapsed
        sol_avg = float(np.mean(sol_ms))
        algo_bytes = ((13 * N + 5) + (10 * N - 4)) * (4 if f32 else 8) + 8
        if lmpc:
            algo_bytes += (7 + 1) * cfgd["num_ss_pts"] * 8  # + ss_x, ss_j in and lambda out (13 936 B at S = 160, 9840 B at S = 96; by reference
            #                                                  the kernel reads 4 S B of codes + the points from the L2-resident store instead, the figure is kept as SURVEY states it)
        achieved = algo_bytes * B / (sol_avg * 1e-3) / 1e9
        # committed PMC passes of this command: tracking, or the learning problem with 160 safe-set points
        pmc_sel = ("r04_pmc_lmpc.json", "lmpc_solve_kernel<double, 4, 3") if lmpc else ("r06_pmc_tracking.json", "lmpc_solve_kernel<double, 4, 0")
        pmc_shape = not (N != 20 or B != 4096 or iac or f32 or mixed)  # the shape the committed passes were taken on
        if lmpc and (args.lmpc_data != "near" or args.laps != 5):        # (r04_pmc_lmpc.json: the rounds-1-4 learning workload, 160 points)
            pmc_shape = False
        traffic, traffic_source, counters = None, None, {}
        if world == 1 and not args.no_pmc:
            wl_argv = ["--batch", str(B), "--horizon", str(N), "--workload", args.workload, "--precision", args.precision, "--lmpc-data", args.lmpc_data, "--ss-mode", args.ss_mode]
            if args.regression:
                wl_argv.append("--regression")
            laps_file = None
            if lmpc:  # the profiled children take the laps of this run (recording them drives ~2000 periods of the tracking loop: minutes under --pmc)
                import tempfile
                laps_file = tempfile.NamedTemporaryFile(suffix=".npz", dir="/tmp", delete=False).name
                np.savez(laps_file, **{"lap%d" % i: np.asarray(a) for i, a in enumerate(laps)})
                wl_argv += ["--laps", str(args.laps), "--laps-npz", laps_file]
            # (SQL LIKE pattern: the learning kernels are the KS = 2 (<= 128 points) / 3 instantiations, the tracking ones KS = 0)
            kname = "lmpc_solve_kernel<%s, %%, %d, " % ("float" if (f32 or mixed) else "double", 0 if not lmpc else (2 if cfgd["num_ss_pts"] <= 128 else 3))
            if not (lmpc or f32 or mixed) and solver.launch_info("f64")["threads_per_problem"] == 128:
                kname = "lmpc_solve_kernel_w2<%"  # the two-wavefronts-per-problem kernels (fp64 tracking, N >= 41: csrc/lmpc_solve_w2.hip.h)
            counters = live_counters(wl_argv, kname, PMC_PASSES[:2] if args.pmc_traffic_only else PMC_PASSES)
            if laps_file:
                os.unlink(laps_file)
            traffic = live_traffic_bytes(counters)
            traffic_source = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two passes of this command at --steps 5, measured in this run"
        if traffic is None and pmc_shape:
            traffic = measured_traffic_bytes(*pmc_sel)
            traffic_source = "profiles/%s (committed rocprofv3 --pmc passes of this command)" % pmc_sel[0]
        engines = live_engines(counters, sol_avg)
        if engines is None and pmc_shape:
            engines = engine_utilisation(sol_avg, *pmc_sel)
        res = {
            "metric": "QP solves/sec (nx=6,nu=2,N=%d)" % N, "value": value, "unit": "solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / timed_steps * 1e3, "higher_is_better": True, "scaling": "weak",
            # the timed region: `steps` x `timed_repeats` steps back to back (>= 200 steps and >= --min-window seconds)
            "timed_steps": timed_steps, "timed_repeats": repeats, "timed_window_s": elapsed,
            "vs_baseline": None, "dtype": "f32" if f32 else ("f32 iteration, f64 arrays" if mixed else "f64"), "data": "synthetic",
            "config": {"workload": (("BARC LMPC with 5-lap safe set (160 points)" if args.laps == 5 else "BARC LMPC with the shipped 3-lap safe set (96 points)") +
                                    ", batch=%d per GPU, N=%d, fp64: safe-set kNN kernel + "
                                    "QP kernel per step (BASELINE configs[2])" + (": laps from the tracking loop at speed scales 0.80 .. 1.0, queries = "
                                    "configs[1]'s random x0 (SURVEY.md 8d config 3)" if args.lmpc_data == "spec" else ": analytic laps, states drawn near the last "
                                    "lap (the workload of rounds 1 - 4; NOT SURVEY.md 8d's)") if lmpc else
                                    ("IAC Putnam tracking MPC, batch=%d per GPU, N=%d, fp32 (BASELINE configs[3])" if f32 else
                                     "IAC Putnam tracking MPC, batch=%d per GPU, N=%d, fp64 (problem of BASELINE configs[3])") if iac else
                                    "BARC tracking MPC, batch=%d random x0 per GPU, N=%d, fp64 (BASELINE configs[1])") % (B, N)
                                   + (" -- mixed precision: fp32 Riccati / interior point between fp64 arrays (BASELINE configs[4])" if mixed else "")
                                   + (" -- error-dynamics regression on: %d recorded sample pairs, every stage regressed before its QP" % len(reg_laps)
                                      if (lmpc and reg_laps) else ""),
                       "batch_per_gpu": B, "horizon": N, "streams": S, "output_layout": args.output_layout,
                       **({"safe_set": "by reference (int32 codes)" if ss_idx_mode else "arrays (ss_x, ss_j)"} if lmpc else {}),
                       "launch_order": ("longest first by the previous solve's iteration counts of the SAME batch (perfect foresight here)"
                                        if orders else "default"), "result_gather": "rccl all_gather (async)" if gather else "none",
                       "ranks_seen": ranks_seen, "gathered": gathered},
            "p50_solve_ms": float(np.percentile(lat, 50)), "p99_solve_ms": float(np.percentile(lat, 99)),
            "latency_samples": len(lat), "value_one_stream": one_stream_value,
            # one batch at a time: the figure that reconciles with kernels_ms and the rocprofv3 summaries (the headline
            # ms_per_step overlaps consecutive batches on `streams` HIP streams, so it can sit below one kernel's duration)
            "ms_per_step_one_stream": (B / one_stream_value * 1e3) if one_stream_value else elapsed / timed_steps * 1e3,
            "batch1_solve_ms": {"p50": float(np.percentile(lat1, 50)), "p99": float(np.percentile(lat1, 99)), "control_period_ms": 25.0} if lat1 else None,
            "solved_fraction": float((st == 0).mean()), "mean_ipm_iters": float(iters.mean()),  # (interior-point iterations + polish rounds)
            "kernels_ms": {"linearize": float(np.mean(lin_ms)), "qp_solve": sol_avg,
                           **({"ss_query": float(np.mean(ss_ms))} if ss_ms else {})},
            "launch": solver.launch_info(args.precision),   # LDS bytes / resident problems per CU of the kernel this entry point launches
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         # counter bytes against algorithmic bytes per launch: what the kernel moves beyond what the problem needs
                         # (spill traffic, the linearisation workspace read back) -- VERDICT r4 item 7
                         "traffic_over_algorithmic": (traffic / (algo_bytes * B)) if traffic else None,
                         "algorithmic_bytes_per_solve": algo_bytes,
                         "note": "algorithmic bytes x batch / lmpc_solve_kernel time; the kernel is "
                                 "FP64-VALU issue / LDS-pipeline bound (DESIGN.md), HBM fraction is reported as required; "
                                 "traffic beyond the algorithmic bytes is the linearisation workspace read back (34 MB per 4096 at N = 20) and "
                                 "register spills -- since round 6 mostly the call frame of the active-set polish (DESIGN.md section 3)",
                         "engines": engines,
                         # the roofline that BINDS this kernel (it is 160 flop/B: HBM is not it) -- instruction issue: busy fractions of the FP64
                         # VALU and of the LDS pipeline over the kernel's duration, and the LDS cycles lost to bank conflicts
                         "issue": ({"valu_busy": engines.get("valu_busy_frac"), "lds_busy": engines.get("lds_busy_frac"),
                                    "bank_conflict_frac": engines.get("lds_bank_conflict_frac"), "source": engines.get("source")} if engines else None)},
        }
        # (the figures that reconcile with profiles/ -- one batch at a time -- also inside `config`, which the driver's parser keeps)
        res["config"]["one_stream"] = {"ms_per_step": res["ms_per_step_one_stream"], "value": one_stream_value, "kernels_ms": res["kernels_ms"]}
        if ss_ms:
            # safe-set query kernel: per query 2 doubles in, 7 S doubles out (ss_x [6][S], ss_j [S]); the lap store
            # (5 laps x ~1320 unrolled points x 2 coordinates) is read once per query from L2, not from HBM
            S_pts = cfgd["num_ss_pts"]
            q_bytes = 2 * 8 + (4 * S_pts if ss_idx_mode else 7 * S_pts * 8)
            t_ss = float(np.mean(ss_ms)) * 1e-3
            res["ss_query_kernel"] = {"ms": t_ss * 1e3, "queries_per_s": B / t_ss, "algorithmic_bytes_per_query": q_bytes,
                                      "achieved_GBps": q_bytes * B / t_ss / 1e9, "frac_of_hbm_peak": q_bytes * B / t_ss / 1e9 / HBM_PEAK_GBS,
                                      "note": "one wave per query: one distance pass per lap (two nearest of each lane's share in registers), bitonic sort of "
                                              "the 64 lane minima, lanes 0..K-1 write the neighbours"}
        if not args.no_cpu_baseline and world == 1:
            try:
                res["cpu_baseline"] = cpu_baseline(pkg, N, B, args.workload, laps=laps if lmpc else None, n_laps=args.laps, budget_s=args.cpu_budget)
            except Exception as ex:  # reported, not hidden
                res["cpu_baseline"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
        default_run = (world == 1 and not lmpc and not iac and not f32 and not mixed and N == 20 and B == 4096)
        if default_run and not args.no_others:
            res["others"] = other_configs(args.steps, args.warmup)
            res["closed_loop"] = closed_loop_leg(pkg, dev)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
