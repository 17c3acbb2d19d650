"""Primary frame of the dragon stand-in at several resolutions on one GPU (device-resident results, HIP events around the
kernel): how the frame time scales once the tail is a smaller share of the frame.  BASELINE config 5 is 3840x2160 over 8 GPUs."""
import ctypes as C
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package()
import torch

sc = pkg.Scene(pkg.scenes.make_dragon(800_000))
L = pkg.lib()
for W, H in ((960, 540), (1920, 1080), (2560, 1440), (3840, 2160), (7680, 4320)):
    cam = pkg.Camera.from_array(pkg.scenes.default_camera(W, H))
    hits = torch.empty(W * H * 16, dtype=torch.uint8, device="cuda")
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(12):
        ev0.record()
        rc = L.cgrt_trace_primary_device(sc._h, C.byref(cam), W, H, 0, 0, W, H, 0, 1, C.c_void_p(hits.data_ptr()), None,
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, L.cgrt_last_error()
        ev1.record()
        torch.cuda.synchronize()
        best = min(best, ev0.elapsed_time(ev1))
    print(f"{W}x{H}: {best:.4f} ms  {W * H / best / 1e3:.0f} Mrays/s", flush=True)
    del hits
