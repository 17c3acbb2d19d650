"""Latency of the per-ray boundary (cgrt_intersect_batch with n = 1, the call BoundingVolumeHierarchy::intersect(Ray&, HitInfo&)
makes) on one thread and on 8 concurrent threads sharing one scene handle, raw ctypes calls (the GIL is released inside).
CGRT_LIB_NAME selects the library: libcgrt_r1.so is round 1's build (3 hipMalloc + 3 hipMemcpy + hipDeviceSynchronize +
3 hipFree per call), libcgrt.so the call-lane version (private stream, persistent scratch, pinned staging)."""
import ctypes as C, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package()
L = C.CDLL(pkg.LIB_PATH)
vp = C.c_void_p
L.cgrt_scene_create.argtypes = [vp, C.c_uint32, vp, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, C.c_int, C.POINTER(vp)]
L.cgrt_intersect_batch.argtypes = [vp, vp, C.c_uint64, vp, vp]
sd = pkg.scenes.SceneData.load(os.path.join(e.ROOT, "tests", "golden", "scenes", "monkey.npz"))
pn = np.ascontiguousarray(sd.pos_nrm, np.float32); tri = np.ascontiguousarray(sd.tri, np.uint32); tm = np.ascontiguousarray(sd.tri_mesh, np.uint32)
mats = np.ascontiguousarray(sd.materials, np.float32)
h = vp()
assert L.cgrt_scene_create(pn.ctypes.data, len(pn), tri.ctypes.data, tm.ctypes.data, len(tri), mats.ctypes.data, len(mats), None, 0, 0, C.byref(h)) == 0
rays = np.zeros((4096, 7), np.float32); rays[:, 0:3] = (1.0, 1.0, -2.5); d = -rays[:, 0:3] + np.random.RandomState(1).uniform(-0.3, 0.3, (4096, 3))
rays[:, 3:6] = d / np.linalg.norm(d, axis=1, keepdims=True); rays[:, 6] = np.finfo(np.float32).max


def loop(k, n, out):
    hit = np.zeros(4, np.uint32); nrm = np.zeros(3, np.float32)
    t0 = time.perf_counter()
    for i in range(n):
        L.cgrt_intersect_batch(h, rays[(k * 997 + i) % 4096].ctypes.data, 1, hit.ctypes.data, nrm.ctypes.data)
    out[k] = (time.perf_counter() - t0) / n * 1e6


o = {}
loop(0, 300, o)  # warm-up
loop(0, 3000, o)
print(f"{os.path.basename(pkg.LIB_PATH)}: 1 thread  {o[0]:8.1f} us per per-ray call")
o = {}
th = [threading.Thread(target=loop, args=(k, 1500, o)) for k in range(8)]
t0 = time.perf_counter()
[t.start() for t in th]; [t.join() for t in th]
wall = time.perf_counter() - t0
print(f"{os.path.basename(pkg.LIB_PATH)}: 8 threads {np.mean(list(o.values())):8.1f} us per call per thread, {8 * 1500 / wall:9.0f} calls/s aggregate")
