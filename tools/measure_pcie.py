"""PCIe-inclusive rate of the host-buffer entry (cgrt_trace_primary: traced into the call lane's device frame, copied back D2H
on every call; a partial frame is seeded H2D with the caller's contents first) next to the HBM-resident rate bench.py reports.
Never used as bench `value`."""
import os, sys, time
import numpy as np
import torch  # (before the library: two HIP runtimes in one process initialise in this order only)
torch.cuda.init()
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
      f"(33.2 MB of CgrtHit D2H into pageable caller memory per call; the traversal itself is ~0.14 ms)")
# pinned caller memory through the device entry: what a host that keeps its frame buffer pinned pays
buf = torch.empty(W * H * 4, dtype=torch.int32, device="cuda"); host = torch.empty(W * H * 4, dtype=torch.int32).pin_memory()
for _ in range(3):
    sc.trace_primary_device(cam, W, H, buf.data_ptr()); host.copy_(buf, non_blocking=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K):
    sc.trace_primary_device(cam, W, H, buf.data_ptr()); host.copy_(buf, non_blocking=True); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"cgrt_trace_primary_device + async D2H into pinned host memory: {dt*1e3:.2f} ms/frame = {W*H/dt/1e6:.1f} Mrays/s")
