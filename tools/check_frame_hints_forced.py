"""Forced frame-hint modes on the larger shares and frames (the policy gives them none): us per frame plain / mode 2 / mode 1."""
import os, sys
import torch

torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e

pkg = e.load_package()
buf = torch.empty(3840 * 2160 * 4, dtype=torch.int32, device="cuda")
scenes = [("dragon 800K", lambda: pkg.scenes.make_dragon(800_000)), ("dragon irregular 800K", lambda: pkg.scenes.make_dragon_irregular(800_000)),
          ("dragon 87K", lambda: pkg.scenes.make_dragon(87_000))]
cases = [("4K share4 r0", 3840, 2160, 0, 4), ("4K share4 r2", 3840, 2160, 2, 4), ("4K share2 r0", 3840, 2160, 0, 2), ("1080p", 1920, 1080, 0, 1), ("1280x720", 1280, 720, 0, 1)]


def t(sc, W, H, rank, n, k=30):
    cam = pkg.scenes.default_camera(W, H)
    f = lambda: sc.trace_primary_device(cam, W, H, buf.data_ptr(), rank=rank, nranks=n)
    for _ in range(14):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / k * 1e3


for name, make in scenes:
    sc = pkg.Scene(make())
    row = []
    for cn, W, H, r, n in cases:
        res = []
        for mode in (0, 2, 1):
            pkg.set_frame_hints(mode)
            res.append(min(t(sc, W, H, r, n) for _ in range(2)))
        row.append(f"{cn} {res[0]:.1f} / {res[1]:.1f} ({(res[1] / res[0] - 1) * 100:+.0f} %) / {res[2]:.1f} ({(res[2] / res[0] - 1) * 100:+.0f} %)")
    print(f"{name}: " + "; ".join(row), flush=True)
    sc.close()
pkg.set_frame_hints(-1)
