"""Secondary-ray lists of the shaded dragon frame, rebuilt on the host (numpy restatement of k_spawn: close enough for timing) and
timed in isolation through cgrt_intersect_batch_device: level-0 shadow rays, mirror rays, level-1 shadow rays."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package()
sd = pkg.scenes.make_dragon(800_000)
W, H = 1920, 1080
cam = pkg.scenes.default_camera(W, H)
sc = pkg.Scene(sd)
rays = sc.generate_rays(cam, W, H)
hits, nrm = sc.intersect(rays)
m = hits["hit"] == 1
o = rays["origin"][m].astype(np.float32); d = rays["direction"][m].astype(np.float32); t = hits["t"][m][:, None]
P = o + d * t
light = sd.point_lights[0, 0:3].astype(np.float32)
def shadow(P):
    tl = light - P; dist = np.linalg.norm(tl, axis=1, keepdims=True).astype(np.float32); dr = (tl / dist).astype(np.float32)
    r = np.zeros(len(P), pkg.RAY_DTYPE); r["origin"] = P + np.float32(1e-3) * dr; r["direction"] = dr; r["t"] = np.finfo(np.float32).max
    return r
def mirror(P, d, n):
    refl = d - 2 * (n * d).sum(1, keepdims=True) * n; refl /= np.linalg.norm(refl, axis=1, keepdims=True)
    r = np.zeros(len(P), pkg.RAY_DTYPE); r["origin"] = P + np.float32(1e-3) * refl; r["direction"] = refl.astype(np.float32); r["t"] = 1.0
    return r
s0 = shadow(P); m0 = mirror(P, d, nrm[m])
h1, n1 = sc.intersect(m0); mm = h1["hit"] == 1
P1 = m0["origin"][mm] + m0["direction"][mm] * h1["t"][mm][:, None]
s1 = shadow(P1)
def timeit(r, n=20):
    dr = torch.from_numpy(r.view(np.float32).reshape(-1, 7).copy()).cuda(); dh = torch.empty(len(r) * 4, dtype=torch.int32, device="cuda")
    for _ in range(3): sc.intersect_device(dr.data_ptr(), len(r), dh.data_ptr())
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(n): sc.intersect_device(dr.data_ptr(), len(r), dh.data_ptr())
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
for name, r in (("shadow L0", s0), ("mirror L0", m0), ("shadow L1", s1), ("shadow L1 x4 (tiled)", np.tile(s1, 4)), ("shadow L1 first 16K", s1[:16384]), ("shadow L1 first 4K", s1[:4096])):
    c = sc.count_batch(r)
    print(f"{name:22s} {len(r):7d} rays  {timeit(r):7.1f} us   nodes/ray {c['sub_visits']/max(1,c['tree_rays']):.1f} tri/ray {c['tri_tests']/max(1,c['tree_rays']):.1f} fallback {c['fallback_rays']}")

# the hardest rays of the first 4K level-1 shadow rays: per-ray work, and their time alone
sub = s1[:4096]
work = np.array([sc.count_batch(sub[i:i + 1])["sub_visits"] for i in range(len(sub))])
order = np.argsort(-work)
print("per-ray node visits: mean %.1f p50 %d p90 %d p99 %d max %d" % (work.mean(), np.percentile(work, 50), np.percentile(work, 90), np.percentile(work, 99), work.max()))
for k in order[:3]:
    c = sc.count_batch(sub[k:k + 1])
    print("ray", int(k), "nodes", c["sub_visits"], "tris", c["tri_tests"], "alone: %.1f us" % timeit(sub[k:k + 1], 10))
top = sub[order[:64]]
print("64 hardest together (one wave): %.1f us;  64 easiest: %.1f us;  ray 0 alone %.1f us" % (timeit(top, 10), timeit(sub[order[-64:]], 10), timeit(sub[0:1], 10)))
sc.set_walk(False)
print("exact walk: 4K list %.1f us, hardest alone %.1f us" % (timeit(sub, 10), timeit(sub[order[0]:order[0] + 1], 10)))
