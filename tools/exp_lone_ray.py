"""Diagnostic: how long do the rays of the bench frame's slowest tile take together (one wave, as in the frame) and alone
(one ray per wave, the other 63 lanes idle)?  The gap is what intra-wave interference costs the longest chains."""
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
rays = sc.generate_rays(cam, W, H).view(np.float32).reshape(H, W, 7)
L = pkg.lib()

def timed(batch):
    dr = torch.from_numpy(np.ascontiguousarray(batch)).cuda()
    dh = torch.empty(len(batch) * 16, dtype=torch.uint8, device="cuda")
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(10):
        ev0.record()
        rc = L.cgrt_intersect_batch_device(sc._h, C.c_void_p(dr.data_ptr()), len(batch), C.c_void_p(dh.data_ptr()), None,
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        ev1.record()
        torch.cuda.synchronize()
        best = min(best, ev0.elapsed_time(ev1))
    return best * 1e3

null = np.array([3.0e38, 3.0e38, 3.0e38, 1, 0, 0, 0], np.float32)
empty = timed(np.tile(null, (64, 1)))
for (x0, y0) in ((904, 504), (1296, 504), (472, 344)):  # tiles of the slowest waves in profiles/r1_step5_wave_anatomy.txt
    tile = rays[y0:y0 + 8, x0:x0 + 8].reshape(64, 7)
    together = timed(tile)
    alone = np.tile(null, (64 * 64, 1))
    alone[::64] = tile
    t_alone = timed(alone)
    per = []
    for k in range(64):
        one = np.tile(null, (64, 1))
        one[0] = tile[k]
        per.append(timed(one))
    per = np.array(per)
    print(f"tile ({x0},{y0}): 64 rays in one wave {together:.1f} us; one ray per wave, 64 waves {t_alone:.1f} us; "
          f"single rays alone: max {per.max():.1f} median {np.median(per):.1f} us; empty launch {empty:.1f} us", flush=True)
