"""Generates tests/golden/scenes/*.npz from the reference's data/*.obj files (run in the dev container,
where /root/reference exists; the GPU box only sees the committed .npz).

The .npz hold the flat arrays cgrt_scene_create takes (vertices as assimp would deliver them, triangles,
mesh ids, materials, the preset's point lights) -- i.e. loaded scene DATA, produced by this repo's own
loader (cg-raytracer_amd/scenes.py); no reference source is involved.
    python tests/golden/make_scene_fixtures.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

DATA = "/root/reference/data"

if __name__ == "__main__":
    pkg = entry.load_package()
    out = os.path.join(ROOT, "tests", "golden", "scenes")
    os.makedirs(out, exist_ok=True)
    for name in ("triangle", "cube", "cornell", "monkey", "dodge"):
        sd = pkg.scenes.load_preset(name, DATA)
        sd.save(os.path.join(out, name + ".npz"))
        print(f"{name}: {sd.ntris} tris, {sd.nmesh} meshes, {len(sd.pos_nrm)} verts")
