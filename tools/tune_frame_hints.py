"""Frame hints: the thresholds (wall time from which a wave's tile counts as hard), swept on the frames the hints are for.
HIP events around 30 back-to-back launches on one stream (frames of one stream run one after the other)."""
import os, sys
import torch

torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e

pkg = e.load_package()
sd = pkg.scenes.make_dragon(800_000)
sc = pkg.Scene(sd)
buf = torch.empty(3840 * 2160 * 4, dtype=torch.int32, device="cuda")


def t(W, H, rank, n, k=30):
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
    return round(a.elapsed_time(b) / k * 1e3, 1)


cases = [("960x540", 960, 540, 0, 1), ("800x800", 800, 800, 0, 1)] + [(f"share8 r{r}", 3840, 2160, r, 8) for r in range(8)]
if os.environ.get("TUNE_MODE1"):
    cases = [(f"share4 r{r}", 3840, 2160, r, 4) for r in range(4)] + [("1080p", 1920, 1080, 0, 1)]
pkg.set_frame_hints(0)
print("no hints:", {n: t(W, H, r, k) for n, W, H, r, k in cases}, flush=True)
for mode in ((1,) if os.environ.get("TUNE_MODE1") else (2,)):
    pkg.set_frame_hints(mode)
    for dense, sparse in ((4000, 2200), (4500, 2500), (4500, 2500), (5000, 2700), (5500, 3000), (3500, 2000)):
        pkg.debug_set_hint_thresholds(dense, sparse)
        print(f"mode {mode}, hard from {dense / 100:.0f} us (64-ray wave) / {sparse / 100:.0f} us (16-ray wave):", {n: t(W, H, r, k) for n, W, H, r, k in cases}, flush=True)
pkg.debug_set_hint_thresholds(0, 0)
pkg.set_frame_hints(-1)
print("auto:", {n: t(W, H, r, k) for n, W, H, r, k in cases}, flush=True)
