"""N > 1 path on CPU: world_size-2 gloo.  The data path has no collective (tiles are independent, the scene
is replicated); what multi-process code there is -- tile ownership, the barrier-bracketed timing and the
MAX-over-ranks reduction bench.py performs -- is exercised here with the same helpers."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, q):
    sys.path.insert(0, ROOT)
    import time

    import torch
    import torch.distributed as dist

    import __graft_entry__ as entry

    entry.load_package()
    from cg_raytracer_amd import tiling

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mask = tiling.owned_mask(W, H, rank, world)
    mine = torch.tensor([int(mask.sum())], dtype=torch.int64)
    # union/disjointness: sum of per-rank ownership == 1 everywhere
    own = torch.from_numpy(mask.astype(np.int32))
    dist.all_reduce(own)
    total = mine.clone()
    dist.all_reduce(total)
    # bench.py's timing protocol: barrier, timed region, barrier, MAX over ranks
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))
    dist.barrier()
    wall = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(wall, op=dist.ReduceOp.MAX)
    q.put((rank, int(mine), int(total), bool((own == 1).all()), float(wall)))
    dist.destroy_process_group()


@pytest.mark.parametrize("W,H", [(1920, 1080), (500, 301)])
def test_two_ranks_partition_and_timing(W, H):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, W, H, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (r0, n0, tot0, ok0, w0), (r1, n1, tot1, ok1, w1) = res
    assert ok0 and ok1 and tot0 == tot1 == W * H and n0 + n1 == W * H
    assert abs(n0 - n1) <= 0.12 * W * H  # interleaved super-tiles roughly balance the pixel count
    assert w0 == w1 and w0 >= 0.1  # both ranks agree on the max-over-ranks time


def test_frame_for_world_is_weak_scaling():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry

    entry.load_package()
    from cg_raytracer_amd import tiling

    assert tiling.frame_for_world(1) == (1920, 1080)
    assert tiling.frame_for_world(4) == (3840, 2160)  # BASELINE.json config 5
    for n in (1, 2, 4, 8):
        w, h = tiling.frame_for_world(n)
        assert w % 8 == 0 and h % 8 == 0
        per_rank = [tiling.owned_pixels(w, h, r, n) for r in range(n)] if n <= 2 else None
        assert abs(w * h / n - 1920 * 1080) / (1920 * 1080) < 0.01
        if per_rank:
            assert sum(per_rank) == w * h


def test_config5_strong_scaling_split():
    """BASELINE config 5 as bench.py --gpus N times it: ONE 3840x2160 frame, super-tile i to rank i % N.  For N = 1, 2, 4, 8
    the ranks own disjoint sets whose union is the frame, and no rank is more than a few super-tiles away from its share."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry

    entry.load_package()
    from cg_raytracer_amd import tiling

    W, H = 3840, 2160
    for n in (1, 2, 4, 8):
        owner = np.full((H, W), -1, np.int32)
        counts = []
        for r in range(n):
            m = tiling.owned_mask(W, H, r, n)
            assert (owner[m] == -1).all()
            owner[m] = r
            counts.append(int(m.sum()))
            assert counts[-1] == tiling.owned_pixels(W, H, r, n)
        assert (owner >= 0).all() and sum(counts) == W * H
        assert max(counts) - min(counts) <= 8 * 64 * 64, counts  # shares differ by a few 64x64 super-tiles at most


def test_bench_cli_documents_the_scaling_modes():
    """bench.py's flags the driver and the docs rely on (no GPU needed for --help)."""
    import subprocess

    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--scaling", "--workload", "--obj", "--tris", "--walk"):
        assert flag in out.stdout
