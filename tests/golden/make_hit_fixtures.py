"""Generates tests/golden/hits_<scene>.npz: seeded rays (tests/rayfam.py) and what the CPU oracle
(oracle/cgrt_oracle.cpp, -O2 build; the -O0 build is asserted identical) returns for them -- hit flag,
t as a u32 bit pattern, primitive id, material index, hitInfo.normal bits, plus the brute-force result
(ray_tracing.cpp:202-213 over all meshes) that exposes the reference's false misses.

The reference has no golden vectors and cannot be built here (DESIGN.md "Oracle"), so
these fixtures pin the ORACLE against drift and give the GPU path committed data to hit; they are not
outputs of the reference binary.
    python tests/golden/make_hit_fixtures.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402
import rayfam  # noqa: E402

SCENES = ["triangle", "cube", "cornell", "monkey", "blob", "spheres"]

if __name__ == "__main__":
    pkg, orc = entry.load_package(), entry.load_oracle()
    out = os.path.join(ROOT, "tests", "golden")
    for name in SCENES:
        if name == "blob":
            sd = pkg.scenes.make_blob(2000, seed=7)
        elif name == "spheres":
            sd = pkg.scenes.spheres_preset()
        else:
            sd = pkg.scenes.SceneData.load(os.path.join(out, "scenes", name + ".npz"))
        o, o0 = orc.OracleScene(sd), orc.OracleScene(sd, o0=True)
        _, boxes = o.nodes()
        W = H = 48
        cam = pkg.scenes.default_camera(W, H)
        fam = rayfam.families(sd, boxes, orc.generate_rays(cam, W, H), n_random=700)
        rays = rayfam.concat(fam)
        a, a0 = o.intersect(rays), o0.intersect(rays)
        assert a.tobytes() == a0.tobytes(), "-O0 and -O2 oracle builds disagree"
        b = o.intersect(rays, brute_force=True)
        np.savez_compressed(
            os.path.join(out, f"hits_{name}.npz"),
            seed=np.uint64(rayfam.SEED), rays=rays,
            hit=a["hit"].astype(np.uint8), t_bits=a["t"].view(np.uint32), prim=a["prim"], material=a["material"],
            normal_bits=a["normal"].view(np.uint32),
            brute_hit=b["hit"].astype(np.uint8), brute_t_bits=b["t"].view(np.uint32), brute_prim=b["prim"],
            levels=np.int32(o.num_levels()),
        )
        dis = int((a["hit"] != b["hit"]).sum())
        print(f"{name}: {len(rays)} rays, {int(a['hit'].sum())} hits, BVH!=brute flags on {dis} rays, levels {o.num_levels()}")
