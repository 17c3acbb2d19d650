"""How much of the bench frame's time is spent on tiles whose rays all fail the root gate?  Times the fused primary kernel on the
whole 1920x1080 frame and on the bounding rectangle (tile-aligned, +1 super-tile margin) of the pixels that can enter the tree."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); orc = e.load_oracle()
sd = pkg.scenes.make_dragon(800_000)
W, H = 1920, 1080
cam = pkg.scenes.default_camera(W, H)
sc = pkg.Scene(sd)
rays = orc.generate_rays(cam, W, H).astype(np.float64)
_, boxes = sc.nodes()
lo, hi = boxes[0, 0:3].astype(np.float64), boxes[0, 3:6].astype(np.float64)
o, d = rays[:, 0:3], rays[:, 3:6]
with np.errstate(divide="ignore", invalid="ignore"):
    t0, t1 = (lo - o) / d, (hi - o) / d
tin, tout = np.minimum(t0, t1).max(1), np.maximum(t0, t1).min(1)
enter = (tin <= tout) & (tout >= 0)
yy, xx = np.divmod(np.nonzero(enter)[0], W)
x0, x1, y0, y1 = max(0, xx.min() // 64 * 64 - 64), min(W, (xx.max() // 64 + 2) * 64), max(0, yy.min() // 64 * 64 - 64), min(H, (yy.max() // 64 + 2) * 64)
print("entering", enter.sum(), "rect", x0, y0, x1, y1, "area share %.3f" % ((x1 - x0) * (y1 - y0) / (W * H)))
tiles = enter.reshape(H // 8, 8, W // 8, 8).any(axis=(1, 3))
print("8x8 tiles with an entering ray: %.3f of all" % tiles.mean())
buf = torch.empty(W * H * 4, dtype=torch.int32, device="cuda")
def timeit(rect, n=50):
    for _ in range(5): sc.trace_primary_device(cam, W, H, buf.data_ptr(), rect=rect)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): sc.trace_primary_device(cam, W, H, buf.data_ptr(), rect=rect)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print("full frame ms", timeit(None), " rect only ms", timeit((int(x0), int(y0), int(x1), int(y1))))
