"""SceneType::CornellBoxSphericalLight (scene.cpp:27-32; BASELINE.md quotes 48.5 s per frame for it, 200 shadow samples per
hit, hardware and resolution unstated) on the device driver, timed by the HIP events inside cgrt_render_soft, with the
oracle's per-pixel driver timed on a band of rows of the same frame beside it.  Also the dragon stand-in under one
spherical light: soft shadows over an 800 K-triangle scene."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as e
pkg = e.load_package()
orc = e.load_oracle()
units = pkg.unit_vector_table(1 << 16, 1)
none = np.zeros((0, 6), np.float32)


def run(name, sd, sl, W, H, depth, samples, cpu_rows):
    cam = pkg.scenes.default_camera(W, H)
    sc = pkg.Scene(sd)
    for closest in (False, True):
        best = None
        for _ in range(3):
            rgb, st = sc.render_soft(cam, W, H, sl, units, samples=samples, seed=1, lights=none, max_level=depth, closest_hit=closest)
            if best is None or st["device_ms"] < best["device_ms"]:
                best = st
        rays = best["primary_rays"] + best["shadow_rays"] + best["reflection_rays"] + best["soft_shadow_rays"]
        print(f"{name} {W}x{H} depth {depth} x{samples} {'closest-hit' if closest else 'any-hit   '}: {best['device_ms']:.2f} ms device, "
              f"{rays} rays ({best['soft_shadow_rays']} soft-shadow samples), {rays / best['device_ms'] / 1e3:.0f} Mrays/s", flush=True)
    if cpu_rows and not os.environ.get("CGRT_NO_CPU"):
        o = orc.OracleScene(sd)
        y0 = H // 2 - cpu_rows // 2
        t0 = time.perf_counter()
        ref, n = o.render_soft(cam, W, H, none, sl, units, samples=samples, seed=1, max_level=depth, y0=y0, y1=y0 + cpu_rows)
        dt = time.perf_counter() - t0
        err = np.abs(ref - rgb.reshape(H, W, 3)[y0:y0 + cpu_rows].reshape(-1, 3)).max()
        print(f"  oracle (CPU, {orc.lib().oracle_max_threads()} threads) rows {y0}..{y0 + cpu_rows}: {dt:.2f} s, {n} rays, {n / dt / 1e6:.1f} Mrays/s "
              f"-> {dt * H / cpu_rows:.1f} s per frame extrapolated; max |RGB diff| vs device on these rows {err:.2e}", flush=True)


cornell = pkg.scenes.SceneData.load(os.path.join(ROOT, "tests/golden/scenes/cornell.npz"))
run("cornell-spherical", cornell, pkg.scenes.CORNELL_SPHERICAL_LIGHTS, 800, 800, 2, 200, 40)
run("cornell-spherical", cornell, pkg.scenes.CORNELL_SPHERICAL_LIGHTS, 1920, 1080, 2, 200, 54)
dragon = pkg.scenes.make_dragon(800_000)
run("dragon800k-spherical", dragon, np.asarray([[-1, 1, -1, 0.1, 1, 1, 1]], np.float32), 1920, 1080, 2, 200, 0)
