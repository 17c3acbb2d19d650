import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as e
pkg = e.load_package(); orc = e.load_oracle()
pkg.set_fast_tree(1)
GOLDEN = os.path.join(os.path.dirname(e.ROOT + "/x"), "tests", "golden")
for name in ("cube", "cornell", "monkey"):
    sd = pkg.scenes.SceneData.load(os.path.join(GOLDEN, "scenes", name + ".npz"))
    sc = pkg.Scene(sd)
    W = H = 128
    cam = pkg.scenes.default_camera(W, H)
    prim = orc.generate_rays(cam, W, H)
    print(name, sc.walk(), sc.count_batch(prim), sc.count_primary(cam, W, H))
for n in (20000, 800000):
    sd = pkg.scenes.make_dragon(n)
    sc = pkg.Scene(sd)
    W, H = 1920, 1080
    cam = pkg.scenes.default_camera(W, H)
    c1 = sc.count_primary(cam, W, H)
    sc.set_walk(False)
    c0 = sc.count_primary(cam, W, H)
    print(n, "certified", c1); print(n, "exact", c0)
