"""Experiment: does re-ordering an incoherent secondary batch pay?  Mirror rays of the dragon frame (one per primary hit,
in tile order as the render driver emits them) traced as they come, sorted by direction octant + Morton code of the
origin, and shuffled (worst case)."""
import ctypes as C
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package()
import torch

sd = pkg.scenes.make_dragon(800_000)
W, H = 1920, 1080
cam = pkg.scenes.default_camera(W, H)
sc = pkg.Scene(sd)
rays = sc.generate_rays(cam, W, H)
h, n = sc.trace_primary(cam, W, H, want_normals=True)
# tile order of the frame (8x8 tiles inside 64x64 super-tiles), like the render driver's lists
ys, xs = np.mgrid[0:H, 0:W]
key = ((ys // 64) * ((W + 63) // 64) + xs // 64) * 4096 + ((ys % 64) // 8 * 8 + (xs % 64) // 8) * 64 + (ys % 8) * 8 + xs % 8
order = np.argsort(key.reshape(-1), kind="stable")
hit = (h["hit"] == 1)[order]
idx = order[hit]
r7 = np.ascontiguousarray(rays.view(np.float32).reshape(-1, 7)[idx])
d = r7[:, 3:6]
nn = n[idx]
p = r7[:, 0:3] + d * h["t"][idx, None]
refl = d - 2.0 * (nn * d).sum(1, keepdims=True) * nn
refl /= np.linalg.norm(refl, axis=1, keepdims=True)
m = np.zeros((len(idx), 7), np.float32)
m[:, 0:3] = p + 0.001 * refl
m[:, 3:6] = refl
m[:, 6] = 3.0e38
print("mirror rays:", len(m), flush=True)

def morton(q):
    q = np.clip(((q + 1.2) / 2.4 * 1023).astype(np.uint64), 0, 1023)
    out = np.zeros(len(q), np.uint64)
    for b in range(10):
        for a in range(3):
            out |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return out

oct_ = ((m[:, 3] < 0).astype(np.uint64) | ((m[:, 4] < 0).astype(np.uint64) << np.uint64(1)) | ((m[:, 5] < 0).astype(np.uint64) << np.uint64(2)))
variants = {"tile order (as emitted)": np.arange(len(m)),
            "octant, then Morton(origin)": np.argsort((oct_ << np.uint64(30)) | morton(m[:, 0:3]), kind="stable"),
            "Morton(origin) only": np.argsort(morton(m[:, 0:3]), kind="stable"),
            "shuffled": np.random.default_rng(1).permutation(len(m))}
L = pkg.lib()
for name, perm in variants.items():
    dr = torch.from_numpy(np.ascontiguousarray(m[perm])).cuda()
    dh = torch.empty(len(m) * 16, dtype=torch.uint8, device="cuda")
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(8):
        ev0.record()
        rc = L.cgrt_intersect_batch_device(sc._h, C.c_void_p(dr.data_ptr()), len(m), C.c_void_p(dh.data_ptr()), None,
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        ev1.record()
        torch.cuda.synchronize()
        best = min(best, ev0.elapsed_time(ev1))
    print(f"{name:32s} {best:.4f} ms  {len(m) / best / 1e3:.0f} Mrays/s", flush=True)
