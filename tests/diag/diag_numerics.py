"""Diagnostic: where do device and oracle bits differ? (run on the GPU box)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as e
pkg = e.load_package(); orc = e.load_oracle()

def ulp(a, b):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    ai = a.view(np.int32).astype(np.int64); bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7fffffff), ai); bi = np.where(bi < 0, -(bi & 0x7fffffff), bi)
    return np.abs(ai - bi)

rng = np.random.RandomState(1)
n = 100000
tri9 = rng.uniform(-1, 1, (n, 9)).astype(np.float32)
a = pkg.triangle_plane(tri9); b = orc.triangle_plane(tri9)
d = ulp(a, b); print("triangle_plane ulp diff per col max", d.max(0), "frac rows differ", (d.max(1) > 0).mean())
sd = pkg.scenes.make_blob(2000, 7)
sc = pkg.Scene(sd)
W = H = 128; cam = pkg.scenes.default_camera(W, H)
r = sc.generate_rays(cam, W, H).view(np.float32).reshape(-1, 7); ro = orc.generate_rays(cam, W, H)
d = ulp(r, ro); print("rays ulp diff per col max", d.max(0), "frac", (d.max(1) > 0).mean())
i = np.argmax(d.max(1)); print(r[i], ro[i])
pin = np.concatenate([tri9, a[:, 1:4], rng.uniform(-1, 1, (n, 3)).astype(np.float32)], 1)
print("pit equal", np.array_equal(pkg.point_in_triangle(pin), orc.point_in_triangle(pin)))
rays = np.zeros((n, 7), np.float32); rays[:, 0:3] = rng.uniform(-2, 2, (n, 3)); dd = rng.normal(size=(n, 3)); rays[:, 3:6] = dd / np.linalg.norm(dd, axis=1, keepdims=True); rays[:, 6] = np.finfo(np.float32).max
t, h = pkg.ray_plane(b, rays.view(pkg.RAY_DTYPE).reshape(-1)); ref = orc.ray_plane(b, rays)
print("ray_plane hit equal", np.array_equal(h, ref["hit"].astype(np.uint8)), "t ulp max", ulp(t, ref["t"]).max(), "frac", (ulp(t, ref["t"]) > 0).mean())
box = np.concatenate([rng.uniform(-1, 0, (n, 3)), rng.uniform(0, 1, (n, 3))], 1).astype(np.float32)
t, h, ins = pkg.ray_box(box, rays.view(pkg.RAY_DTYPE).reshape(-1)); ref = orc.ray_box(box, rays)
print("ray_box hit equal", np.array_equal(h, ref["hit"].astype(np.uint8)), "t ulp max", ulp(t, ref["t"]).max(), "frac", (ulp(t, ref["t"]) > 0).mean())
