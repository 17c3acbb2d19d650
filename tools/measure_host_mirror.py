"""The C++ host mirror's two render drivers on the Cornell box at 1920x1080 (wall time of the call, scene creation excluded is
not possible through this entry: the BVH is built per call, so the driver's own seconds are what is compared)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sd = pkg.scenes.SceneData.load(os.path.join(ROOT, "tests/golden/scenes/cornell.npz"))
W, H = 1920, 1080
cam = pkg.scenes.default_camera(W, H)
none = np.zeros((0, 7), np.float32)
for on_device in (False, True, False, True):
    rgb, st = pkg.host_render_soft(sd, cam, W, H, none, None, max_level=2, on_device=on_device)
    print(f"on_device={on_device}: driver total {st['seconds_total'] * 1e3:.2f} ms, of which device/batch calls {st['seconds_device'] * 1e3:.2f} ms; "
          f"rays {st['primary'] + st['shadow'] + st['reflection']}", flush=True)
t = pkg.host_time_screen_render(sd, cam, W, H, max_level=2, reps=5)
print(f"Screen-filling drivers, whole call, BVH built once, best of 5: renderRayTracingOnDevice {t['on_device_ms']:.2f} ms "
      f"(device share {t['on_device_device_share_ms']:.3f} ms; pinned frame -> Screen::setFrame), renderRayTracing (host-driven wavefront) "
      f"{t['host_wavefront_ms']:.1f} ms")
