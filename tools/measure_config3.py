"""BASELINE.json config 3 (CornellBox-Mirror-Rotated 1920x1080, recursion depth 4) on the device wavefront, and the
dragon stand-in with full shading, timed by the HIP events inside cgrt_render (all kernels of the frame)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package()
W, H = 1920, 1080
cam = pkg.scenes.default_camera(W, H)
for name, sd, depth in (("cornell", pkg.scenes.SceneData.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/scenes/cornell.npz")), 4),
                        ("cornell", pkg.scenes.SceneData.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/scenes/cornell.npz")), 2),
                        ("dragon800k", pkg.scenes.make_dragon(800_000), 2)):
    if len(sys.argv) > 1 and sys.argv[1] != name:
        continue
    sc = pkg.Scene(sd)
    best = None
    for _ in range(5):
        rgb, st = sc.render(cam, W, H, max_level=depth)
        if best is None or st["device_ms"] < best["device_ms"]:
            best = st
    rays = best["primary_rays"] + best["shadow_rays"] + best["reflection_rays"]
    print(f"{name} {W}x{H} depth {depth}: {best['device_ms']:.3f} ms device, {rays} rays "
          f"({best['primary_rays']} primary, {best['shadow_rays']} shadow, {best['reflection_rays']} mirror), "
          f"{rays / best['device_ms'] / 1e3:.1f} Mrays/s, levels {best['levels']}")
