import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as e
pkg = e.load_package(); orc = e.load_oracle()
import test_adversarial_gpu as T
for log2_scale in (38, 39):
    rng = np.random.RandomState(77 + log2_scale)
    base = T._sliver_scene(pkg, rng, ntris=1500)
    s = np.float32(2.0) ** np.float32(log2_scale)
    pn = base.pos_nrm.copy(); pn[:, 0:3] = (pn[:, 0:3] * s).astype(np.float32)
    sd = pkg.scenes.SceneData(pos_nrm=pn, tri=base.tri, tri_mesh=base.tri_mesh, materials=base.materials)
    rays = T._aimed_rays(base, rng, n=1200); rays[:, 0:6] = (rays[:, 0:6] * s).astype(np.float32)
    ref = orc.OracleScene(sd).intersect(rays)
    sc = pkg.Scene(sd)
    print("scale", log2_scale, "walk", sc.walk(), "subnodes", sc.num_subnodes())
    R = rays.view(pkg.RAY_DTYPE).reshape(-1)
    h1, _ = sc.intersect(R)
    sc.set_walk(False); h0, _ = sc.intersect(R)
    pkg.set_leaf_accel(False); lin = pkg.Scene(sd); pkg.set_leaf_accel(True)
    hl, _ = lin.intersect(R)
    for name, h in (("certified", h1), ("exact+accel", h0), ("linear", hl)):
        bad = np.nonzero((h["hit"] != ref["hit"]) | (h["t"].view(np.uint32) != ref["t"].view(np.uint32)))[0]
        print(" ", name, "bad", len(bad), bad[:5])
        for i in bad[:3]:
            print("    ray", rays[i].tolist(), "got", h[i], "ref", (ref["hit"][i], ref["t"][i], ref["prim"][i]))
            p = int(ref["prim"][i]) if ref["hit"][i] else int(h["prim_id"][i])
            if p < sd.ntris:
                v = sd.pos_nrm[sd.tri[p], 0:3]; print("    tri", p, v.tolist())
