#!/usr/bin/env python3
"""bench.py -- primary Mrays/s + achieved algorithmic GB/s of the BVH traversal + ray/triangle path on the
dragon stand-in at 1920x1080 (BASELINE.json metric), one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (fused primary-ray generation + BoundingVolumeHierarchy::intersect for
every pixel the rank owns) with the scene resident in HBM.  N > 1: the frame grows with N (weak scaling,
tiling.frame_for_world), 8x8 tiles are dealt round-robin to ranks, no data-path collective; the timed
region is bracketed by barrier + synchronize and the MAX over ranks is taken.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(pkg, sd, cam, W, H, budget_s=15.0):
    """The oracle (CPU restatement of the reference algorithm, kind "port") timed on this box's host cores
    on a bounded sample of the same frame: a centred block of rows sized for ~budget_s of work, run as one
    `omp parallel for` over rows exactly like main.cpp:653-656, wall time by std::chrono like :791-797."""
    orc = entry.load_oracle()
    threads = os.cpu_count() or 1
    t0 = time.time()
    o = orc.OracleScene(sd)
    build_s = time.time() - t0
    probe = min(H, 2 * threads)
    y0 = (H - probe) // 2
    s, _ = o.trace_primary_timed(cam, W, H, y0=y0, y1=y0 + probe, threads=threads)
    blk = int(max(probe, min(H, probe * budget_s / max(s, 1e-6))))
    y0 = (H - blk) // 2
    s, _ = o.trace_primary_timed(cam, W, H, y0=y0, y1=y0 + blk, threads=threads)
    return {
        "value": round(blk * W / s / 1e6, 4),
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "sample": f"rows {y0}..{y0 + blk} ({blk * W} primary rays) of the same {W}x{H} frame in {s:.1f}s, "
                  f"g++ -O2 oracle, omp parallel for over rows; oracle BVH build {build_s:.1f}s not included",
    }


def read_traffic(workload: str):
    """HBM bytes per launch from the committed PMC pass (profiles/hbm_traffic_latest.json, written by
    tools/profile_round.sh on the GPU box: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950
    correction applied as MI355X_MICROARCH.md prescribes).  None when no pass matches this workload."""
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic_latest.json")
    try:
        tj = json.load(open(tpath))
        if tj.get("workload") == workload:
            return tj.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--tris", type=int, default=800_000, help="triangles of the procedural dragon stand-in")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--frames-in-flight", type=int, default=1,
                    help="independent frames kept in flight on separate HIP streams (1 = strictly back to back, the reported default)")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run --nproc-per-node N (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one rank per GPU; CGRT_BENCH_BACKEND=gloo lets several ranks share one card to rehearse the N>1 path on a 1-GPU box
    backend = os.environ.get("CGRT_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    pkg = entry.load_package()
    from cg_raytracer_amd import tiling

    W, H = tiling.frame_for_world(world)
    if args.width and args.height:
        W, H = args.width, args.height
    if os.environ.get("CGRT_PRIMARY_MODE"):  # experiment knob: 0 = wave per tile, 1 = persistent waves with lane refill
        pkg.set_primary_mode(int(os.environ["CGRT_PRIMARY_MODE"]))
    if os.environ.get("CGRT_SUB_LEAF"):  # experiment knob: triangles per in-leaf accelerator run
        pkg.set_leaf_accel(True, int(os.environ["CGRT_SUB_LEAF"]))
    sd = pkg.scenes.make_dragon(args.tris)
    cam = pkg.scenes.default_camera(W, H)
    t0 = time.time()
    scene = pkg.Scene(sd, device=local_rank)
    setup_s = time.time() - t0

    nfl = max(1, args.frames_in_flight)
    hits_bufs = [torch.empty(W * H * 4, dtype=torch.int32, device="cuda") for _ in range(nfl)]  # CgrtHit x W*H each, in HBM
    hits = hits_bufs[0]
    main_stream = torch.cuda.current_stream()
    streams = [main_stream] + [torch.cuda.Stream() for _ in range(nfl - 1)]
    stream = main_stream.cuda_stream
    step_no = [0]

    def step():
        k = step_no[0] % nfl
        step_no[0] += 1
        scene.trace_primary_device(cam, W, H, hits_bufs[k].data_ptr(), rank=rank, nranks=world, stream=streams[k].cuda_stream)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_start = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    for st in streams[1:]:
        main_stream.wait_stream(st)  # the closing event covers every frame in flight
    ev1.record()
    barrier()
    wall = time.perf_counter() - t_start
    kern_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream: avg launch duration

    my_rays = tiling.owned_pixels(W, H, rank, world)
    cnt = scene.count_primary(cam, W, H, rank=rank, nranks=world)  # instrumented launch, outside the timed region
    assert cnt["rays"] == my_rays, (cnt, my_rays)
    rs = pkg.record_sizes()
    alg_bytes = cnt["inner_visits"] * rs["node"] + cnt["tri_tests"] * rs["tri"] + cnt["sub_visits"] * rs["sub"] + cnt["rays"] * rs["hit"]

    wall_t = torch.tensor([wall], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall_max = float(wall_t.item())
    total_rays = W * H
    if rank == 0:
        ms_per_step = wall_max / args.steps * 1e3
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic = read_traffic(f"dragon{sd.ntris}_{W}x{H}")
        out = {
            "metric": "primary Mrays/sec (BVH traversal + ray-triangle, dragon stand-in @1920x1080 per GPU)",
            "value": round(total_rays / (wall_max / args.steps) / 1e6, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (procedural dragon stand-in: data/dragon.obj is absent from the reference checkout)",
            "config": {
                "workload": f"dragon stand-in {sd.ntris} tris (1 mesh, 12-level reference BVH), {W}x{H} primary rays, "
                            f"reference default camera, 8x8 tiles interleaved over {world} rank(s)",
                "rays_per_step": total_rays,
                "frames_in_flight": nfl,
                "rays_rank0": my_rays,
                "bvh_build_and_upload_s": round(setup_s, 3),
                "scene_device_MB": round(scene.device_bytes() / 1e6, 1),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "kernel": "k_trace_primary",
                "kernel_ms": round(kern_ms, 4),
                "algorithmic_bytes_per_launch": int(alg_bytes),
                "per_ray": {k: round(cnt[k] / max(1, cnt["rays"]), 2) for k in ("inner_visits", "leaf_visits", "tri_tests", "sub_visits")},
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pkg, sd, cam, W, H, args.cpu_budget)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
