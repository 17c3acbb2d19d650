import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package()
W, H = 1920, 1080
cam = pkg.scenes.default_camera(W, H)
root = os.path.join(e.ROOT, "tests/golden/scenes")
for mode in (-1, 1):
    pkg.set_fast_tree(mode)
    for name, depth in (("cornell", 4), ("monkey", 2), ("cube", 2)):
        sd = pkg.scenes.SceneData.load(os.path.join(root, name + ".npz"))
        sc = pkg.Scene(sd)
        best = min(sc.render(cam, W, H, max_level=depth)[1]["device_ms"] for _ in range(6))
        c = sc.count_primary(cam, W, H)
        print(f"fast_tree={mode} {name} depth {depth}: {best:.3f} ms; walk {sc.walk()} primary tree rays {c['tree_rays']} fallback {c['fallback_rays']} inner {c['inner_visits']} sub {c['sub_visits']}")
