#!/usr/bin/env python3
"""bench.py -- primary Mrays/s + achieved algorithmic GB/s of the BVH traversal + ray/triangle path on the
dragon stand-in (BASELINE.json metric), one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W                       # headline: 1920x1080 (BASELINE config 4)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
                                                                       # BASELINE config 5: ONE 3840x2160 frame split over N ranks

A step = one pass of the hot path (fused primary-ray generation + BoundingVolumeHierarchy::intersect for every pixel
the rank owns) with the scene resident in HBM.  The frame is cut into 64x64-pixel super-tiles, super-tile i belongs to
rank i % N, the scene is replicated, no data-path collective exists; the timed region is bracketed by barrier +
synchronize and the MAX over ranks is taken.  Rank 0 prints ONE JSON line.

Scaling modes (N > 1): "strong" (default) = the fixed 3840x2160 frame of BASELINE.json's config 5 -- the reference's one
parallel construct is rows of one frame (main.cpp:653-656); "weak" (--scaling weak) = the frame grows with N so that
per-GPU work stays at 1920x1080.  `--scaling strong` at N = 1 renders the config-5 frame on one GPU (the base of the curve).
Other workloads: --workload shaded (cgrt_render: primary + shadow + mirror rays, the reference's whole frame),
--tris 87000 (the report's dragon size), --obj PATH (a user-supplied mesh, e.g. the real dragon.obj).
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
CONFIG5 = (3840, 2160)  # BASELINE.json configs[4]


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cpus():
    """CPUs this process may actually use: the cgroup's quota when /sys/fs/cgroup/cpu.max states one (the GPU box shows 256
    hardware threads and grants 16 CPUs), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def frame_hint_note(mode, rays_per_rank, world):
    """What cgrt_set_frame_hints does to this launch (the library's policy restated: capi.cpp hint_mode_for)."""
    eff = mode
    if mode < 0:
        eff = 2 if rays_per_rank <= 800_000 or (world >= 4 and rays_per_rank <= 1_300_000) else 0
    what = {0: "none (plain launch order)",
            1: "the tiles whose wave took long in the previous frame (>= 45 us, rising while more than 2 % of the tiles qualify) are traced first",
            2: "the tiles whose wave took long in the previous frame (>= 45 us, rising while more than 2 % of the tiles qualify) are traced as four 16-ray waves, first"}[eff]
    return (f"cgrt_set_frame_hints({mode}): {what}" + ("" if eff == 0 else "; the first frame of a shape is plain; same pixels either way "
            "(tests/test_frame_hints_gpu.py, profiles/r3_frame_hints.txt)"))


def cpu_baseline(sd, cam, W, H, budget_s=12.0):
    """The oracle (CPU restatement of the reference algorithm, kind "port") timed on this box's host cores on a bounded
    sample of the same frame: a centred block of rows, run as one `omp parallel for` over rows exactly like
    main.cpp:653-656, wall time by std::chrono like :791-797.  Two builds, as SURVEY.md section 8(d) asks: -O2 and -O0
    (the reference's de-facto build type, CMake sets none); each: one warm-up, then best of 3.  `value` is the -O2 figure."""
    orc = entry.load_oracle()
    threads = host_cpus()
    legs = {}
    sample = ""
    for o0 in (False, True):
        t0 = time.time()
        o = orc.OracleScene(sd, o0=o0)
        build_s = time.time() - t0
        probe = min(H, 2 * threads)
        y0 = (H - probe) // 2
        s, _ = o.trace_primary_timed(cam, W, H, y0=y0, y1=y0 + probe, threads=threads)  # warm-up + sizing probe
        per_leg = budget_s / 2 / 3  # three timed repetitions per build inside the budget
        blk = int(max(probe, min(H, probe * per_leg / max(s, 1e-6))))
        y0 = (H - blk) // 2
        best = min(o.trace_primary_timed(cam, W, H, y0=y0, y1=y0 + blk, threads=threads)[0] for _ in range(3))
        legs["O0" if o0 else "O2"] = {"Mrays_per_s": round(blk * W / best / 1e6, 4), "rows": blk, "best_of_3_s": round(best, 3),
                                      "oracle_bvh_build_s": round(build_s, 2)}
        if not o0:
            sample = (f"rows {y0}..{y0 + blk} ({blk * W} primary rays) of the same {W}x{H} frame, omp parallel for over rows, "
                      f"one warm-up then best of 3; -O0 leg sized the same way")
        o.close()
    return {
        "value": legs["O2"]["Mrays_per_s"],
        "unit": "Mrays/s",
        "cores": threads,
        "cpu_model": cpu_model(),
        "kind": "port",
        "sample": sample,
        "builds": legs,
    }


def read_profile(workload: str, lib_hash: str):
    """What the committed PMC passes say about the dominant kernel (profiles/hbm_traffic_latest.json, written by
    tools/profile_round.sh on the GPU box: rocprofv3 --pmc in separate passes, corrected as MI355X_MICROARCH.md prescribes):
    HBM bytes per launch and the `measured` fractions that show what bounds the kernel.  NOT measured by this run, and only
    valid for the kernels they were measured on: the file records the source hash of the library it profiled
    (cgrt_source_hash) and everything is nulled when that differs from the library this run loaded, or when the workload
    differs.  Returns (traffic, measured, source note)."""
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic_latest.json")
    try:
        tj = json.load(open(tpath))
    except Exception:
        return None, None, "no committed PMC pass"
    if tj.get("workload") != workload:
        return None, None, f"profiles/hbm_traffic_latest.json is for {tj.get('workload')}, not this workload"
    if tj.get("source_hash") != lib_hash:
        return None, None, (f"profiles/hbm_traffic_latest.json was measured on sources {tj.get('source_hash')}, this library is built from {lib_hash}: "
                            "stale, not reported (re-run tools/profile_round.sh)")
    return tj.get("hbm_bytes_per_launch"), tj.get("measured"), f"profiles/hbm_traffic_latest.json ({tj.get('source', 'PMC pass')}; sources {lib_hash}), not measured by this run"


def bound_from(measured, hbm_physical_frac):
    """Names what the counters show as the limit.  SURVEY.md 8(d) classifies the path as HBM-bound; the committed counters of the
    dominant kernel decide what is printed."""
    if not measured:
        return "hbm"
    if hbm_physical_frac is not None and hbm_physical_frac >= 0.5:
        return "hbm"
    if measured.get("valu_issue_frac", 0) >= 0.75:
        return "valu-issue"
    return "latency (vector-L1 queueing + dependent VALU chains; not HBM)"


def algorithmic_bytes(cnt, rs):
    """record bytes of every visit + the result store; 24 B per certificate box."""
    return (cnt["inner_visits"] * rs["node"] + cnt["tri_tests"] * rs["tri"] + cnt["sub_visits"] * rs["sub"] + cnt["cert_boxes"] * 24
            + cnt["rays"] * rs["hit"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--tris", type=int, default=800_000, help="triangles of the procedural dragon stand-in (87000 = the report's size)")
    ap.add_argument("--obj", type=str, default="", help="load this OBJ (centred + unit-scaled like scene.cpp:42) instead of the stand-in")
    ap.add_argument("--standin", choices=["regular", "irregular"], default="regular",
                    help="regular = the tessellated torus-knot tube (default); irregular = the same shape with scan-like density, slivers, noise and order")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--scaling", choices=["auto", "strong", "weak"], default="auto",
                    help="auto: N = 1 -> the 1920x1080 headline frame, N > 1 -> strong (fixed 3840x2160 frame, BASELINE config 5)")
    ap.add_argument("--frame-hints", type=int, default=-1, choices=[-1, 0, 1, 2],
                    help="cgrt_set_frame_hints: -1 the library's policy by frame size (default), 0 off, 1 hard tiles first, 2 hard tiles as 16-ray waves")
    ap.add_argument("--workload", choices=["primary", "shaded"], default="primary",
                    help="primary = fused ray generation + intersect (the headline); shaded = cgrt_render, depth 2, all rays of the frame")
    ap.add_argument("--walk", choices=["auto", "exact", "certified"], default="auto",
                    help="auto = the scene's default (certified when it carries a fast tree), exact = the reference's steps only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary lines (config-5 frame on one GPU, 87 K dragon) at N = 1")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
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
    pkg.set_frame_hints(args.frame_hints)
    from cg_raytracer_amd import tiling

    scaling = args.scaling if args.scaling != "auto" else ("strong" if world > 1 else "headline")
    if scaling == "strong":
        W, H = CONFIG5
    elif scaling == "weak":
        W, H = tiling.frame_for_world(world)
    else:
        W, H = tiling.BASE_W, tiling.BASE_H
    if args.width and args.height:
        W, H = args.width, args.height
    if os.environ.get("CGRT_PRIMARY_MODE"):  # experiment knob: 0 = wave per tile, 1 = persistent waves with lane refill
        pkg.set_primary_mode(int(os.environ["CGRT_PRIMARY_MODE"]))
    if os.environ.get("CGRT_SUB_LEAF"):  # experiment knob: triangles per in-leaf accelerator run
        pkg.set_leaf_accel(True, int(os.environ["CGRT_SUB_LEAF"]))
    if args.obj:
        sd = pkg.host_load_obj(args.obj, normalize=True)
        scene_name = f"{os.path.basename(args.obj)} ({sd.ntris} tris, {sd.nmesh} mesh(es))"
        data = f"user-supplied OBJ {os.path.basename(args.obj)}"
    else:
        sd = pkg.scenes.make_dragon(args.tris) if args.standin == "regular" else pkg.scenes.make_dragon_irregular(args.tris)
        scene_name = f"dragon stand-in {sd.ntris} tris (1 mesh, 12-level reference BVH)" + ("" if args.standin == "regular" else ", irregular variant")
        data = "synthetic (procedural dragon stand-in: data/dragon.obj is absent from the reference checkout)"
    cam = pkg.scenes.default_camera(W, H)
    t0 = time.time()
    scene = pkg.Scene(sd, device=local_rank)
    setup_s = time.time() - t0
    if args.walk != "auto":
        scene.set_walk(args.walk == "certified")
    if os.environ.get("CGRT_PRIMARY_MODE") == "1":
        # the persistent-waves variant (variant_kernels.hip) walks exactly, whatever the scene carries: the line, its counters and
        # its traffic key must describe what was launched
        scene.set_walk(False)
    walk = "certified (fast tree + certificate, exact fallback)" if scene.walk() else "exact"
    if os.environ.get("CGRT_PRIMARY_MODE") == "1":
        walk += ", persistent waves with lane refill (cgrt_set_primary_mode(1))"

    main_stream = torch.cuda.current_stream()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step_fn, steps, warmup, streams=()):
        """W untimed steps, then exactly K steps between barrier + synchronize; returns (wall seconds, event ms per step)."""
        for _ in range(warmup):
            step_fn()
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t_start = time.perf_counter()
        ev0.record()
        for _ in range(steps):
            step_fn()
        for st in streams:
            main_stream.wait_stream(st)  # the closing event covers every frame in flight
        ev1.record()
        barrier()
        return time.perf_counter() - t_start, ev0.elapsed_time(ev1) / steps  # HIP events on the launch stream

    def primary_runner(sc, cam_, W_, H_, rank_, world_, nfl=1):
        bufs = [torch.empty(W_ * H_ * 4, dtype=torch.int32, device="cuda") for _ in range(nfl)]  # CgrtHit x W*H each, in HBM
        streams = [main_stream] + [torch.cuda.Stream() for _ in range(nfl - 1)]
        k = [0]

        def step():
            i = k[0] % nfl
            k[0] += 1
            sc.trace_primary_device(cam_, W_, H_, bufs[i].data_ptr(), rank=rank_, nranks=world_, stream=streams[i].cuda_stream)

        return step, streams[1:], bufs

    rs = pkg.record_sizes()
    nfl = max(1, args.frames_in_flight)
    out = None
    if args.workload == "primary":
        step, extra_streams, bufs = primary_runner(scene, cam, W, H, rank, world, nfl)
        wall, kern_ms = timed(step, args.steps, args.warmup, extra_streams)
        my_rays = tiling.owned_pixels(W, H, rank, world)
        cnt = scene.count_primary(cam, W, H, rank=rank, nranks=world)  # instrumented launch, outside the timed region
        assert cnt["rays"] == my_rays, (cnt, my_rays)
        alg_bytes = algorithmic_bytes(cnt, rs)
        # the gather of SURVEY.md section 8(e): this rank's hits device -> pinned host, one async copy, timed on its own
        host = torch.empty(W * H * 4, dtype=torch.int32).pin_memory()
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        host.copy_(bufs[0], non_blocking=True)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        per_rank = [my_rays]
        walls = torch.tensor([wall, gather_ms], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        if dist is not None:
            dist.all_reduce(walls, op=dist.ReduceOp.MAX)
            lst = [None] * world
            dist.all_gather_object(lst, my_rays)
            per_rank = [int(x) for x in lst]
        wall_max, gather_max = float(walls[0].item()), float(walls[1].item())
        total_rays = W * H
        n1_same = None
        if world > 1 and rank == 0 and scaling != "weak":
            # the base of the strong-scaling curve, measured by rank 0 alone on the whole frame after the timed region
            s1, _, _ = primary_runner(scene, cam, W, H, 0, 1)
            for _ in range(3):
                s1()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                s1()
            torch.cuda.synchronize()
            n1_same = total_rays / ((time.perf_counter() - t1) / 10) / 1e6
        if rank == 0:
            ms_per_step = wall_max / args.steps * 1e3
            achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
            traffic, measured, traffic_src = read_profile(f"dragon{sd.ntris}_{W}x{H}_{'certified' if scene.walk() else 'exact'}", pkg.source_hash())
            phys_frac = round(traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None
            tree = max(1, cnt["tree_rays"])
            out = {
                "metric": (f"primary Mrays/sec (BVH traversal + ray-triangle, {'user OBJ' if args.obj else 'dragon stand-in'} @{W}x{H}"
                           + (f", one frame split over {world} GPUs, strong scaling" if world > 1 and scaling == "strong" else
                              (f", frame grown with {world} GPUs, weak scaling" if world > 1 else "")) + ")"),
                "value": round(total_rays / (wall_max / args.steps) / 1e6, 3),
                "unit": "Mrays/s",
                "n_gpus": world,
                "steps": args.steps,
                "warmup": args.warmup,
                "ms_per_step": round(ms_per_step, 4),
                "higher_is_better": True,
                "vs_baseline": None,
                "dtype": "f32",
                "data": data,
                "config": {
                    "workload": f"{scene_name}, {W}x{H} primary rays, reference default camera, 64x64-pixel super-tiles dealt i % {world} to "
                                f"{world} rank(s), {walk} walk",
                    "rays_per_step": total_rays,
                    "rays_entering_tree_rank0": cnt["tree_rays"],
                    "Mrays_per_s_over_rays_entering_tree_rank0": round(cnt["tree_rays"] / (kern_ms * 1e-3) / 1e6, 1),
                    "frames_in_flight": nfl,
                    "frame_hints": frame_hint_note(args.frame_hints, per_rank[0] if isinstance(per_rank, (list, tuple)) else per_rank, world),
                    "rays_per_rank": per_rank,
                    "gather_ms_max_over_ranks": round(gather_max, 3),
                    "gather": "this rank's CgrtHit frame device -> pinned host, one async copy (not part of value)",
                    "bvh_build_and_upload_s": round(setup_s, 3),
                    "scene_device_MB": round(scene.device_bytes() / 1e6, 1),
                    "timing": "value and ms_per_step: wall clock between barriers (max over ranks) / steps; roofline.kernel_ms: HIP events on the launch stream (rank 0)",
                },
                "roofline": {
                    "bound": bound_from(measured, phys_frac),
                    "bound_note": "SURVEY.md 8(d) classifies the path as HBM-bound and defines frac = algorithmic bytes (every record visit + the result "
                                  "store) / kernel time / 8 TB/s; the records are re-read through L1/L2/Infinity Cache, physical HBM traffic is a few % of "
                                  "that (hbm_physical_frac), and the algorithmic fraction EXCEEDS 1 on the 3840x2160 frame on one GPU (see "
                                  "extras.config5_frame_on_one_gpu.algorithmic_frac): it is not a bound for this kernel.  `bound` and `measured` say what "
                                  "the committed PMC passes show (valu_issue_frac = SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles, l1_busy_frac = TCP_GATE_EN1 / "
                                  "CU-cycles, waitcnt_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES, l1_cycles_per_load = TCP_TCP_LATENCY / TCP_TA_TCP_STATE_READ)",
                    "achieved": round(achieved, 2),
                    "peak": HBM_PEAK_GBS,
                    "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": traffic,
                    "traffic_source": traffic_src,
                    "hbm_physical_frac": phys_frac,
                    "measured": measured,
                    "source_hash": pkg.source_hash(),
                    "kernel": "k_trace_primary",
                    "kernel_ms": round(kern_ms, 4),
                    "algorithmic_bytes_per_launch": int(alg_bytes),
                    "algorithmic_bytes_per_ray_entering_tree": round((alg_bytes - cnt["rays"] * rs["hit"]) / tree + rs["hit"], 1),
                    "per_ray": {k: round(cnt[k] / max(1, cnt["rays"]), 3) for k in ("inner_visits", "leaf_visits", "tri_tests", "sub_visits", "cert_boxes")},
                    "per_ray_entering_tree": {k: round(cnt[k] / tree, 2) for k in ("inner_visits", "tri_tests", "sub_visits", "cert_boxes")},
                    "fallback_rays": cnt["fallback_rays"],
                },
            }
            if world > 1:
                out["scaling"] = scaling
                if n1_same is not None:
                    out["config"]["n1_same_frame_Mrays_per_s"] = round(n1_same, 1)
                    out["config"]["n1_same_frame_note"] = "rank 0 alone on the whole frame, after the timed region: the base of this curve"
    else:  # shaded: the reference's whole frame (renderRayTracing, main.cpp:648-720) on the device wavefront, depth 2
        if world > 1:
            raise SystemExit("--workload shaded is a single-GPU line")
        best = None
        for _ in range(args.warmup):
            scene.render(cam, W, H, max_level=2)
        t_start = time.perf_counter()
        for _ in range(args.steps):
            _, st = scene.render(cam, W, H, max_level=2)
            if best is None or st["device_ms"] < best["device_ms"]:
                best = st
        wall = time.perf_counter() - t_start
        rays = best["primary_rays"] + best["shadow_rays"] + best["reflection_rays"]
        path = scene.last_render_path()
        # the same frame drawn the exact way (what a first frame, a resized frame or a frame after a camera jump costs): after the timed region
        pkg.set_render_prediction(False)
        exact_ms = min(scene.render(cam, W, H, max_level=2)[1]["device_ms"] for _ in range(max(3, args.steps // 4)))
        pkg.set_render_prediction(True)
        _, _, work = scene.render_counted(cam, W, H, max_level=2)  # instrumented frame, outside the timed region
        parts = {k: algorithmic_bytes(v, rs) for k, v in work.items()}
        alg_bytes = sum(parts.values())
        achieved = alg_bytes / (best["device_ms"] * 1e-3) / 1e9
        out = {
            "metric": f"rays/sec of the whole shaded frame (primary + shadow + mirror, depth 2, {scene_name} @{W}x{H})",
            "value": round(rays / (best["device_ms"] * 1e-3) / 1e6, 3),
            "unit": "Mrays/s",
            "n_gpus": 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(best["device_ms"], 4),
            "higher_is_better": True,
            "vs_baseline": None,
            "dtype": "f32",
            "data": data,
            "config": {"workload": f"{scene_name}, {W}x{H}, cgrt_render depth 2 ({walk} walk): {best['primary_rays']} primary, "
                                   f"{best['shadow_rays']} shadow, {best['reflection_rays']} mirror rays",
                       "timing": "ms_per_step: HIP events around all kernels of the best frame (device side of cgrt_render); "
                                 f"host-inclusive mean {wall / args.steps * 1e3:.3f} ms (RGB download included)",
                       "frame_path": {0: "exactly sized (the host waits for the device's hit count inside the frame)",
                                      1: "predicted: every launch issued at once, sized from the previous frame of the same shape, counts checked behind the frame "
                                         "(cgrt_set_render_prediction; same pixels, tests/test_render_prediction_gpu.py)",
                                      2: "predicted, outgrown, drawn again exactly"}.get(path, str(path)),
                       "exactly_sized_frame_ms": round(exact_ms, 4)},
            "roofline": {
                "bound": "hbm",
                "bound_note": "classification of SURVEY.md 8(d); like the primary kernel these batches are latency-bound (their time is the time of their hardest rays), see profiles/",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None,
                "kernel": "all traversal kernels of the frame (k_trace_primary_compact, k_trace_shadow, k_trace_batch) over the frame's device time, shading kernels included in the time",
                "kernel_ms": round(best["device_ms"], 4),
                "algorithmic_bytes_per_launch": int(alg_bytes),
                "algorithmic_bytes_by_ray_kind": {k: int(v) for k, v in parts.items()},
                "per_ray_entering_tree": {kind: {k: round(v[k] / max(1, v["tree_rays"]), 2) for k in ("inner_visits", "tri_tests", "sub_visits", "cert_boxes")}
                                          for kind, v in work.items()},
                "rays_entering_tree": {kind: v["tree_rays"] for kind, v in work.items()},
                "fallback_rays": {kind: v["fallback_rays"] for kind, v in work.items()},
            },
        }
    if rank == 0 and out is not None:
        if world == 1 and args.workload == "primary" and not args.no_extras and scaling == "headline" and not args.obj and args.standin == "regular" and not (args.width and args.height):
            # secondary lines of SURVEY.md section 8(d): the config-5 frame on this one GPU (base of the strong-scaling curve)
            # and the report's 87 K-triangle dragon beside the 800 K one
            extras = {}
            ksteps = max(10, args.steps // 2)

            def line(sc_, sd_, cam_, W_, H_):
                """one secondary line: frame time, rate, counters and the algorithmic fraction of the same definition as roofline.frac"""
                st, _, _ = primary_runner(sc_, cam_, W_, H_, 0, 1)
                w_, k_ = timed(st, ksteps, 3)
                c_ = sc_.count_primary(cam_, W_, H_)
                tr = max(1, c_["tree_rays"])
                return {"tris": sd_.ntris, "frame": f"{W_}x{H_}", "Mrays_per_s": round(W_ * H_ / (w_ / ksteps) / 1e6, 1), "kernel_ms": round(k_, 4),
                        "rays_entering_tree": c_["tree_rays"], "Mrays_per_s_over_rays_entering_tree": round(c_["tree_rays"] / (k_ * 1e-3) / 1e6, 1),
                        "per_ray_entering_tree": {k: round(c_[k] / tr, 2) for k in ("inner_visits", "tri_tests", "sub_visits", "cert_boxes")},
                        "fallback_rays": c_["fallback_rays"],
                        "algorithmic_frac": round(algorithmic_bytes(c_, rs) / (k_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "walk": "certified" if sc_.walk() else "exact"}

            W5, H5 = CONFIG5
            extras["config5_frame_on_one_gpu"] = line(scene, sd, pkg.scenes.default_camera(W5, H5), W5, H5)
            extras["config5_frame_on_one_gpu"]["note"] = "BASELINE config 5's frame on ONE GPU: the base of the strong-scaling curve; algorithmic_frac > 1 here"
            if args.tris != 87_000:
                sd87 = pkg.scenes.make_dragon(87_000)
                sc87 = pkg.Scene(sd87, device=local_rank)
                if args.walk != "auto":
                    sc87.set_walk(args.walk == "certified")
                extras["dragon_87k_report_size"] = line(sc87, sd87, cam, W, H)
                sc87.close()
            # less flattering scenes: the irregular stand-in (scan-like density, slivers, noise, shuffled order) and
            # the largest real mesh the reference ships (data/dodgeColorTest.obj: 16 311 triangles, 11 meshes; committed as arrays)
            sdi = pkg.scenes.make_dragon_irregular(args.tris)
            sci = pkg.Scene(sdi, device=local_rank)
            if args.walk != "auto":
                sci.set_walk(args.walk == "certified")
            extras["dragon_irregular_standin"] = line(sci, sdi, cam, W, H)
            sci.close()
            dodge_path = os.path.join(ROOT, "tests", "golden", "scenes", "dodge.npz")
            if os.path.exists(dodge_path):
                sdd = pkg.scenes.SceneData.load(dodge_path)
                scd = pkg.Scene(sdd, device=local_rank)
                extras["dodgeColorTest_obj"] = line(scd, sdd, cam, W, H)
                extras["dodgeColorTest_obj"]["meshes"] = int(sdd.nmesh)
                extras["dodgeColorTest_obj"]["source"] = "reference data/dodgeColorTest.obj (the largest real mesh it ships), arrays committed as tests/golden/scenes/dodge.npz"
                scd.close()
            # the latency regime: small frames, and the slowest-looking share of config 5 split over 8 GPUs, traced
            # on this one card -- without frame hints and with the library's default hints (cgrt_set_frame_hints; same pixels).  The
            # hints' threshold settles over the first frames of a shape: 10 warm-up frames, then the timed ones.
            def small(W_, H_, rank_, n_, mode):
                pkg.set_frame_hints(mode)
                st, _, _ = primary_runner(scene, pkg.scenes.default_camera(W_, H_), W_, H_, rank_, n_)
                w_, k_ = timed(st, ksteps, 10)
                return round(k_ * 1e3, 1)

            lat = {}
            for name, (W_, H_, r_, n_) in {"640x360": (640, 360, 0, 1), "960x540": (960, 540, 0, 1),
                                           "one_eighth_of_3840x2160_rank0": (CONFIG5[0], CONFIG5[1], 0, 8)}.items():
                lat[name] = {"kernel_us_no_hints": small(W_, H_, r_, n_, 0), "kernel_us_default_hints": small(W_, H_, r_, n_, -1)}
            pkg.set_frame_hints(args.frame_hints)
            lat["note"] = ("HIP events around back-to-back frames of one shape on one stream, dragon stand-in; default hints = the tiles whose wave "
                           "was long in the previous frame are traced as four 16-ray waves (DESIGN.md 5.8, profiles/r3_frame_hints.txt)")
            extras["latency_regime"] = lat
            out["extras"] = extras
        if not args.no_cpu_baseline:
            # N > 1: rank 0 alone, after the timed region's closing barrier (the other ranks wait at the final barrier below), so that
            # every line of the 1/2/4/8 run carries the CPU reference "in the same run" (BASELINE.json north_star)
            out["cpu_baseline"] = cpu_baseline(sd, cam, W, H, args.cpu_budget)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
