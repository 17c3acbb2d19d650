"""Frame hints, frame by frame: device time of every frame of a sequence (HIP events around each launch) and the lengths of the
three rotating hard lists afterwards.  Usage: python tools/diag_frame_hints.py W H rank nranks mode [dense_ticks sparse_ticks]"""
import os, sys
import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e

pkg = e.load_package()
W, H, rank, nranks, mode = (int(v) for v in sys.argv[1:6])
if len(sys.argv) > 7:
    pkg.debug_set_hint_thresholds(int(sys.argv[6]), int(sys.argv[7]))
sd = pkg.scenes.make_dragon(800_000)
sc = pkg.Scene(sd)
cam = pkg.scenes.default_camera(W, H)
buf = torch.empty(W * H * 4, dtype=torch.int32, device="cuda")


def frames(n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for k in range(n):
        sc.trace_primary_device(cam, W, H, buf.data_ptr(), rank=rank, nranks=nranks)
        ev[k + 1].record()
    torch.cuda.synchronize()
    return [round(ev[k].elapsed_time(ev[k + 1]) * 1e3, 1) for k in range(n)]


pkg.set_frame_hints(0)
frames(5)
print(f"{W}x{H} rank {rank}/{nranks}: no hints   ", frames(12))
pkg.set_frame_hints(mode)
print(f"mode {mode}: first frames ", frames(12), "lists", sc.hint_counts())
print(f"mode {mode}: later frames ", frames(12), "lists", sc.hint_counts())
t = frames(60)
print(f"mode {mode}: mean of 60 {np.mean(t):.1f} us, min {min(t)}, max {max(t)}; lists", sc.hint_counts())
