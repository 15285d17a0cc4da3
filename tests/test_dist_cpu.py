"""The N > 1 path of bench.py on CPU: world_size-2 gloo, shard + result gather (no GPU, no kernels).

The hot path shards by problem with no data-path collective; the only exchange is the gather of
results.  This test drives the same shard/pack/all_gather/unpack helpers bench.py uses, with the
solve replaced by a deterministic stand-in (the product has no CPU solver by design)."""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def _worker(rank, world, port, N, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench

    lo, hi = bench.shard_bounds(world * B, world, rank)
    assert hi - lo == B
    # stand-in "results": value encodes (global problem index, field, element)
    idx = torch.arange(lo, hi, dtype=torch.float64)
    out = {"X_optm": idx[None, None, :] + torch.arange(6 * N, dtype=torch.float64).reshape(6, N, 1) * 1e-3,
           "U_optm": -idx[None, None, :] + torch.arange(2 * (N - 1), dtype=torch.float64).reshape(2, N - 1, 1) * 1e-3,
           "dU_optm": 2 * idx[None, None, :] + torch.zeros(2, N - 1, 1, dtype=torch.float64),
           "status": (torch.arange(lo, hi) % 3).to(torch.int32), "iters": (7 + torch.arange(lo, hi)).to(torch.int32)}
    flat = torch.empty(bench.packed_numel(N, B), dtype=torch.float64)
    gbuf = torch.empty(world * flat.numel(), dtype=torch.float64)
    bench.pack_results(out, flat)
    h = dist.all_gather_into_tensor(gbuf, flat, async_op=True)
    h.wait()
    full = bench.unpack_results(gbuf, world, N, B)
    if rank == 0:
        q.put({k: v.numpy() for k, v in full.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    N, B, world = 6, 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    full = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert full["X_optm"].shape == (6, N, world * B)
    gi = np.arange(world * B, dtype=np.float64)
    assert np.array_equal(full["X_optm"][0, 0], gi)
    assert np.allclose(full["X_optm"][5, N - 1], gi + (6 * N - 1) * 1e-3)
    assert np.array_equal(full["U_optm"][0, 0], -gi)
    assert np.array_equal(full["dU_optm"][1, N - 2], 2 * gi)
    # status and iteration counts travel with the results: rank 0 knows which problems of which rank failed
    assert full["status"].dtype == np.int32 and np.array_equal(full["status"], np.arange(world * B) % 3)
    assert np.array_equal(full["iters"], 7 + np.arange(world * B))


def test_shard_bounds_cover_everything():
    import bench

    for total, world in ((4096, 8), (10, 3), (7, 8)):
        cuts = [bench.shard_bounds(total, world, r) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == total
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
