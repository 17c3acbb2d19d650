"""One-off check at BASELINE config 5's frame size on ONE GPU: dragon stand-in 800 K triangles, 3840x2160 primary rays, every
pixel against the oracle (flag, t bits, primitive id, material, normal bits), plus the 2-rank tiling merged."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as e
pkg = e.load_package()
orc = e.load_oracle()
sd = pkg.scenes.make_dragon(800_000)
W, H = 3840, 2160
cam = pkg.scenes.default_camera(W, H)
sc = pkg.Scene(sd)
t0 = time.time()
h, n = sc.trace_primary(cam, W, H, want_normals=True)
print("device frame incl. transfers %.2f s" % (time.time() - t0), flush=True)
o = orc.OracleScene(sd)
t0 = time.time()
ref = o.intersect(orc.generate_rays(cam, W, H))
print("oracle %.1f s" % (time.time() - t0), flush=True)
hit = ref["hit"] == 1
same_t = (h["t"].view(np.uint32) == ref["t"].view(np.uint32)) | (np.isnan(h["t"]) & np.isnan(ref["t"]))
ok = np.array_equal(h["hit"], ref["hit"]) and same_t.all() and np.array_equal(h["prim_id"], ref["prim"]) and \
    np.array_equal(h["material_id"], ref["material"])
okn = np.array_equal(np.ascontiguousarray(n[hit]).view(np.uint32), np.ascontiguousarray(ref["normal"][hit]).view(np.uint32))
print("4K frame: hits", int(hit.sum()), "flag/t/prim/material identical:", bool(ok), " normals identical:", bool(okn), flush=True)
m = np.full_like(h, 0)
m["t"] = np.nan
for r in range(2):
    hr, _ = sc.trace_primary(cam, W, H, rank=r, nranks=2)
    own = pkg.tiling.owned_mask(W, H, r, 2).reshape(-1)
    m[own] = hr[own]
print("2-rank merge identical to the single-rank frame:", bool(m.tobytes() == h.tobytes()), flush=True)
