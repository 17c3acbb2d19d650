"""The latency-regime figures VERDICT r2 item 1 names, measured in one process on one scene build (dragon stand-in, 800 K):
  share8   slowest 1/8 share of the 3840x2160 frame (what an 8-GPU node's step would be; tools/predict_strong_scaling.py)
  list4k   the first 4 096 level-1 shadow rays of the shaded frame, through cgrt_intersect_batch_device
  list64k  the whole level-1 shadow list
  lone     the hardest of those rays alone
  q540     960x540 primary frame;  p1080 the headline frame;  p2160 the config-5 frame on one GPU
  shaded   cgrt_render depth 2 at 1920x1080 (device ms of the best of 10 frames)
HIP events around K back-to-back launches (each launch is longer than the host's launch path, so the figure is device time).
Usage: [CGRT_LIB_NAME=libcgrt_x.so] python tools/latency_suite.py [tag]"""
import json, os, sys
import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e

pkg = e.load_package()
tag = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("CGRT_LIB_NAME", "libcgrt.so")
K = 20
sd = pkg.scenes.make_dragon(800_000)
sc = pkg.Scene(sd)


def ev_time(fn, k=K, warm=8):  # (frame hints are set up at a shape's third frame and settle over the next few)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / k * 1e3  # us


out = {"tag": tag}
PARTS = os.environ.get("SUITE_PARTS", "frames,lists,shaded").split(",")
buf = torch.empty(3840 * 2160 * 4, dtype=torch.int32, device="cuda")


def frame(W, H, rank=0, n=1):
    cam = pkg.scenes.default_camera(W, H)
    return ev_time(lambda: sc.trace_primary_device(cam, W, H, buf.data_ptr(), rank=rank, nranks=n))


if "frames" in PARTS:
    out["share8_us"] = [round(frame(3840, 2160, r, 8), 1) for r in range(8)]
    out["share8_max_us"] = max(out["share8_us"])
    out["share4_max_us"] = round(max(frame(3840, 2160, r, 4) for r in range(4)), 1)
    out["p2160_us"] = round(frame(3840, 2160), 1)
    out["p1080_us"] = round(frame(1920, 1080), 1)
    out["q540_us"] = round(frame(960, 540), 1)
if "lists" not in PARTS and "shaded" not in PARTS:
    print(json.dumps(out), flush=True)
    sys.exit(0)

# the shaded frame's level-1 shadow list, rebuilt on the host (as tools/exp_secondary_lists.py)
W, H = 1920, 1080
cam = pkg.scenes.default_camera(W, H)
rays = sc.generate_rays(cam, W, H)
hits, nrm = sc.intersect(rays)
m = hits["hit"] == 1
o = rays["origin"][m].astype(np.float32)
d = rays["direction"][m].astype(np.float32)
P = o + d * hits["t"][m][:, None]
light = sd.point_lights[0, 0:3].astype(np.float32)


def shadow(P):
    tl = light - P
    dist = np.linalg.norm(tl, axis=1, keepdims=True).astype(np.float32)
    dr = (tl / dist).astype(np.float32)
    r = np.zeros(len(P), pkg.RAY_DTYPE)
    r["origin"] = P + np.float32(1e-3) * dr
    r["direction"] = dr
    r["t"] = np.finfo(np.float32).max
    return r


def mirror(P, d, n):
    refl = d - 2 * (n * d).sum(1, keepdims=True) * n
    refl /= np.linalg.norm(refl, axis=1, keepdims=True)
    r = np.zeros(len(P), pkg.RAY_DTYPE)
    r["origin"] = P + np.float32(1e-3) * refl
    r["direction"] = refl.astype(np.float32)
    r["t"] = 1.0
    return r


m0 = mirror(P, d, nrm[m])
h1, _ = sc.intersect(m0)
mm = h1["hit"] == 1
P1 = m0["origin"][mm] + m0["direction"][mm] * h1["t"][mm][:, None]
s1 = shadow(P1)
s0 = shadow(P)


def list_time(r, k=K):
    dr = torch.from_numpy(r.view(np.float32).reshape(-1, 7).copy()).cuda()
    dh = torch.empty(len(r) * 4, dtype=torch.int32, device="cuda")
    return round(ev_time(lambda: sc.intersect_device(dr.data_ptr(), len(r), dh.data_ptr()), k), 1)


sub = s1[:4096]
if "lists" not in PARTS:
    best = min(sc.render(cam, W, H, max_level=2)[1]["device_ms"] for _ in range(10))
    out["shaded_ms"] = round(best, 4)
    print(json.dumps(out), flush=True)
    sys.exit(0)
out["list4k_us"] = list_time(sub)
out["list16k_us"] = list_time(s1[:16384])
out["list64k_us"] = list_time(s1)
out["list64k_n"] = len(s1)
out["shadow_l0_us"] = list_time(s0)
out["shadow_l0_n"] = len(s0)
out["mirror_l0_us"] = list_time(m0)
# the hardest ray of the 4K list alone (by node visits of the certified walk), and one wave of the 64 hardest
work = np.array([sc.count_batch(sub[i:i + 1])["sub_visits"] for i in range(0, len(sub), 1)]) if os.environ.get("SUITE_LONE", "1") == "1" else None
if work is not None:
    order = np.argsort(-work)
    out["lone_nodes"] = int(work[order[0]])
    out["lone_us"] = list_time(sub[order[0]:order[0] + 1], 10)
    out["hard64_us"] = list_time(sub[order[:64]], 10)
    out["easy64_us"] = list_time(sub[order[-64:]], 10)
# miss-only launch: rays that fail the root gate (launch + setup cost alone)
miss = np.zeros(4096, pkg.RAY_DTYPE)
miss["origin"] = (10, 10, 10)
miss["direction"] = (0, 0, 1)
miss["t"] = np.finfo(np.float32).max
out["miss4k_us"] = list_time(miss)
if "shaded" in PARTS:
    best = min(sc.render(cam, W, H, max_level=2)[1]["device_ms"] for _ in range(10))
    out["shaded_ms"] = round(best, 4)
print(json.dumps(out), flush=True)
