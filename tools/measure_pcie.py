"""PCIe-inclusive rate of the host-buffer entry (cgrt_trace_primary: result buffer allocated, seeded H2D, traced,
copied back D2H on every call) next to the HBM-resident rate bench.py reports.  Never used as bench `value`."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package()
sd = pkg.scenes.make_dragon(800_000)
W, H = 1920, 1080
cam = pkg.scenes.default_camera(W, H)
sc = pkg.Scene(sd)
for _ in range(3):
    sc.trace_primary(cam, W, H)
t0 = time.perf_counter(); K = 10
for _ in range(K):
    sc.trace_primary(cam, W, H)
dt = (time.perf_counter() - t0) / K
print(f"cgrt_trace_primary (host buffers, {W}x{H}, {sd.ntris} tris): {dt*1e3:.2f} ms/frame = {W*H/dt/1e6:.1f} Mrays/s "
      f"(33.2 MB of CgrtHit H2D seed + 33.2 MB D2H + hipMalloc/hipFree per call; the traversal itself is ~0.3 ms)")
