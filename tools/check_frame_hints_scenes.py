"""Frame hints on scenes other than the one they were tuned on: us per frame without / with the library's default hints."""
import os, sys
import torch

torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e

pkg = e.load_package()
buf = torch.empty(3840 * 2160 * 4, dtype=torch.int32, device="cuda")
G = os.path.join(e.ROOT, "tests", "golden", "scenes")
scenes = [("dragon 800K", lambda: pkg.scenes.make_dragon(800_000)), ("dragon irregular 800K", lambda: pkg.scenes.make_dragon_irregular(800_000)),
          ("dragon 87K", lambda: pkg.scenes.make_dragon(87_000)), ("dodge", lambda: pkg.scenes.SceneData.load(os.path.join(G, "dodge.npz"))),
          ("monkey", lambda: pkg.scenes.SceneData.load(os.path.join(G, "monkey.npz"))), ("cornell", lambda: pkg.scenes.SceneData.load(os.path.join(G, "cornell.npz")))]
cases = [("640x360", 640, 360, 0, 1), ("800x800", 800, 800, 0, 1), ("960x540", 960, 540, 0, 1), ("1280x720", 1280, 720, 0, 1), ("4K share8 r0", 3840, 2160, 0, 8), ("4K share8 r3", 3840, 2160, 3, 8),
         ("4K share4 r0", 3840, 2160, 0, 4), ("1080p share2 r0", 1920, 1080, 0, 2)]


def t(sc, W, H, rank, n, k=30):
    cam = pkg.scenes.default_camera(W, H)
    f = lambda: sc.trace_primary_device(cam, W, H, buf.data_ptr(), rank=rank, nranks=n)
    for _ in range(6):
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
        pkg.set_frame_hints(0)
        p = min(t(sc, W, H, r, n) for _ in range(2))
        pkg.set_frame_hints(-1)
        h = min(t(sc, W, H, r, n) for _ in range(2))
        row.append(f"{cn} {p:.1f} -> {h:.1f} ({(h / p - 1) * 100:+.0f} %)")
    print(f"{name}: " + "; ".join(row), flush=True)
    sc.close()
