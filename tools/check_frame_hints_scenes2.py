"""Frame hints on the irregular stand-in and others: list lengths and times for several thresholds."""
import os, sys
import torch

torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e

pkg = e.load_package()
buf = torch.empty(3840 * 2160 * 4, dtype=torch.int32, device="cuda")
G = os.path.join(e.ROOT, "tests", "golden", "scenes")
scenes = [("dragon 800K", lambda: pkg.scenes.make_dragon(800_000)), ("dragon irregular 800K", lambda: pkg.scenes.make_dragon_irregular(800_000)),
          ("dragon 87K", lambda: pkg.scenes.make_dragon(87_000)), ("monkey", lambda: pkg.scenes.SceneData.load(os.path.join(G, "monkey.npz")))]
cases = [("800x800", 800, 800, 0, 1), ("960x540", 960, 540, 0, 1), ("4K share8 r0", 3840, 2160, 0, 8), ("1080p share2 r0", 1920, 1080, 0, 2)]


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
    for cn, W, H, r, n in cases:
        pkg.set_frame_hints(0)
        p = min(t(sc, W, H, r, n) for _ in range(2))
        tiles = ((W + 7) // 8) * ((H + 7) // 8) // n
        out = []
        for thr in ((4500, 2500), (7000, 3800), (10000, 5500)):
            pkg.debug_set_hint_thresholds(*thr)
            pkg.set_frame_hints(2)
            h = min(t(sc, W, H, r, n) for _ in range(2))
            cnt = max(sc.hint_counts())
            out.append(f"{thr[0] // 100} us: {h:.1f} us, {cnt} listed = {100.0 * cnt / tiles:.1f} % of the tiles")
        pkg.debug_set_hint_thresholds(0, 0)
        print(f"{name} {cn}: plain {p:.1f} us, {tiles} tiles; mode 2 from " + "; ".join(out), flush=True)
    sc.close()
