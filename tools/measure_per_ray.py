"""The UNCHANGED host at a usable speed?  Two measurements on the GPU box, printed for INTEGRATION.md:
 1. aggregate per-ray call rate -- T std::threads calling BoundingVolumeHierarchy::intersect(Ray&, HitInfo&) on ONE object through the
    C++ mirror (cgrt_host_threads_test), call combining off (round 2: a launch per ray) and on, T = 1, 8, 64, 256;
 2. the reference's frame through its own structure taken literally (`render --per-ray`: omp parallel for over rows, per-pixel
    recursive getFinalColor, one intersect call per ray): monkey 800x800 (windowResolution, main.cpp:29), depth 2, with T caller
    threads -- beside the CPU oracle's time for the SAME frame (recursive per-pixel driver, all host cores, -O2 and -O0) and the
    mirror's batched wavefront / device driver."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e

import bench

pkg = e.load_package()
orc = e.load_oracle()
CPUS = bench.host_cpus()  # the cgroup's CPU quota (16 on the GPU box, which shows 256 hardware threads)
sd = pkg.scenes.SceneData.load(os.path.join(e.ROOT, "tests", "golden", "scenes", "monkey.npz"))
W = H = 800
cam = pkg.scenes.default_camera(W, H)
rays = orc.generate_rays(cam, W, H)
print(f"host: {os.cpu_count()} hardware threads, {CPUS} usable (cgroup quota); scene monkey-rotated.obj ({sd.ntris} triangles), {W}x{H}")
QUICK = os.environ.get("PER_RAY_QUICK") == "1"  # only the call-rate table with combining on


def throttled_ms():
    """Milliseconds this cgroup has spent throttled by its CPU quota so far (cgroup v2 cpu.stat), or None."""
    try:
        for ln in open("/sys/fs/cgroup/cpu.stat"):
            if ln.startswith("throttled_usec"):
                return int(ln.split()[1]) / 1e3
    except OSError:
        pass
    return None


for combining in ((True,) if QUICK else (False, True)):
    pkg.set_call_combining(combining)
    for T in (1, 8, 64, 256):
        n = min(len(rays), 4000 * T if combining else 1500 * T)
        thr0, t0 = throttled_ms(), time.time()
        bad, tim = pkg.host_threads_test(sd, rays[:n], nthreads=T)
        thr1, t1 = throttled_ms(), time.time()
        print(f"call combining {'on ' if combining else 'off'} {T:4d} threads: {tim['calls_per_second']:10.0f} per-ray calls/s "
              f"({tim['us_per_call_per_thread']:7.1f} us per call per thread; one thread alone {tim['us_per_call_one_thread']:.1f} us), disagreements {bad}"
              + (f"; {tim['combined_rays'] / max(1, tim['combined_generations']):.1f} rays per launch on average (largest {tim['largest_generation']}), "
                 f"{tim['leader_gpu_us_per_generation']:.1f} us launch + kernel + wait per launch, {tim['leader_launch_us_per_generation']:.1f} of them up to the launch call's return" if combining else "")
              + (f"; cgroup throttled {thr1 - thr0:.0f} CPU-ms during the test's {1e3 * (t1 - t0):.0f} ms" if thr0 is not None else ""), flush=True)
pkg.set_call_combining(True)
if QUICK:
    sys.exit(0)
o2 = orc.OracleScene(sd)
t0 = time.time()
ref, nrays = o2.render(cam, W, H, sd.point_lights, max_level=2, threads=CPUS)
t_o2 = time.time() - t0
o0 = orc.OracleScene(sd, o0=True)
t0 = time.time()
o0.render(cam, W, H, sd.point_lights, max_level=2, threads=CPUS)
t_o0 = time.time() - t0
print(f"frame {W}x{H} depth 2 = {nrays} rays: CPU oracle (recursive per-pixel driver, {CPUS} threads) -O2 {t_o2 * 1e3:.0f} ms, -O0 {t_o0 * 1e3:.0f} ms "
      f"(BASELINE.md: the report's monkey frame 0.5 s on unstated hardware)")
for T in (8, 64, 256):
    rgb, st = pkg.host_render_per_ray(sd, cam, W, H, 2, threads=T)
    err = np.abs(rgb.astype(np.float64) - ref).max()
    print(f"render --per-ray, {T:3d} caller threads: {st['seconds_total'] * 1e3:8.0f} ms ({nrays / st['seconds_total'] / 1e6:.2f} M per-ray calls/s), max |dRGB| vs oracle {err:.2e}", flush=True)
pkg.set_call_combining(False)
rgb, st = pkg.host_render_per_ray(sd, cam, W, H, 2, threads=8)
print(f"render --per-ray,   8 caller threads, combining OFF (round 2's boundary): {st['seconds_total'] * 1e3:8.0f} ms")
pkg.set_call_combining(True)
rgb, st = pkg.host_render(sd, cam, W, H, 2)
print(f"mirror's batched wavefront (renderRayTracing): {st['seconds_total'] * 1e3:.1f} ms")
sc = pkg.Scene(sd)
sc.render(cam, W, H, max_level=2)
t0 = time.time()
_, st = sc.render(cam, W, H, max_level=2)
print(f"device driver (cgrt_render): {(time.time() - t0) * 1e3:.1f} ms whole call, {st['device_ms']:.3f} ms on the device")
