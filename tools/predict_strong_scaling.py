"""BASELINE config 5's strong-scaling curve, predicted on ONE card: the 3840x2160 frame is split over N ranks exactly as
`bench.py --gpus N` splits it (64x64 super-tiles, tile i -> rank i % N), and every rank's share is timed on this GPU, one
share at a time.  A rank's share on its own GPU costs what it costs here (replicated scene, no exchange), so
max over ranks = the step time an N-GPU node would show, minus launch skew and the barrier.  This is a prediction from
measured shares, not a measurement of N GPUs; it is labelled so wherever it is quoted."""
import json, os, sys, time
import torch  # (before the library: two HIP runtimes in one process initialise in this order only)
torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package()
from cg_raytracer_amd import tiling

W, H = 3840, 2160
K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
HINTS = int(os.environ.get("PREDICT_FRAME_HINTS", "-1"))  # cgrt_set_frame_hints: -1 = the library's default policy, 0 = off
pkg.set_frame_hints(HINTS)
sd = pkg.scenes.make_dragon(800_000)
cam = pkg.scenes.default_camera(W, H)
sc = pkg.Scene(sd)
buf = torch.empty(W * H * 4, dtype=torch.int32, device="cuda")
rows = []
for n in (1, 2, 4, 8):
    shares = []
    for r in range(n):
        for _ in range(8):  # (frame hints are set up at a shape's third frame and settle over the next few)
            sc.trace_primary_device(cam, W, H, buf.data_ptr(), rank=r, nranks=n)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(K):
            sc.trace_primary_device(cam, W, H, buf.data_ptr(), rank=r, nranks=n)
        ev1.record()
        torch.cuda.synchronize()
        cnt = sc.count_primary(cam, W, H, rank=r, nranks=n)
        shares.append({"rank": r, "rays": tiling.owned_pixels(W, H, r, n), "tree_rays": cnt["tree_rays"], "ms": round(ev0.elapsed_time(ev1) / K, 4)})
    worst = max(s["ms"] for s in shares)
    rows.append({"n_gpus": n, "predicted_ms_per_step": worst, "predicted_Mrays_per_s": round(W * H / worst / 1e3, 1), "shares": shares})
base = rows[0]["predicted_ms_per_step"]
for row in rows:
    row["predicted_speedup"] = round(base / row["predicted_ms_per_step"], 3)
    row["predicted_efficiency"] = round(base / row["predicted_ms_per_step"] / row["n_gpus"], 3)
print(json.dumps({"what": "config 5 (3840x2160, 800K-triangle dragon stand-in) strong scaling PREDICTED from per-rank shares timed one at a time on one MI355X",
                  "steps_per_share": K,
                  "frame_hints": {-1: "library default (cgrt_set_frame_hints -1): a share's frames after the first trace their hard tiles first (<= 2.6 M rays) "
                                      "or as 16-ray waves (<= 1.3 M rays), from the previous frame's per-wave times; same pixels", 0: "off"}.get(HINTS, str(HINTS)),
                  "curve": rows}, indent=1))
