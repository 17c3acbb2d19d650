"""Render fuzz (run on the GPU box): random scenes, cameras, point lights and recursion depths through cgrt_render (the whole
shading / recursion driver on the device, certified walk + occlusion queries where the scene has a fast tree) against the
oracle's recursive per-pixel driver: RGB within 1e-5 (BASELINE.json's tolerance), ray counts equal, and the exact-walk render
of the same frame byte-identical.  Since round 3 certified frames are also rendered in the quad-per-ray kernel shape (byte-identical).
Usage: python tools/fuzz_render.py [seconds] [seed]; tests/test_fuzz_gpu.py calls run() over the committed seed list."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as e
pkg = e.load_package(); orc = e.load_oracle()


def run(budget=120.0, seed0=1, verbose=True):
    """Fuzz for `budget` seconds from seed `seed0`; returns the statistics (bad_rgb, bad_counts, walks_differ must be 0)."""
    t_end = time.time() + budget
    stats = dict(frames=0, rays=0, certified_frames=0, max_err=0.0, bad_rgb=0, bad_counts=0, walks_differ=0, shapes_differ=0)
    t_last = time.time()
    it = 0
    while time.time() < t_end:
        rng = np.random.RandomState(seed0 * 7919 + it); it += 1
        kind = rng.randint(0, 4)
        if kind == 0: sd = pkg.scenes.make_dragon(int(rng.choice([2000, 12000, 60000])), seed=int(rng.randint(1, 1 << 30)))
        elif kind == 1: sd = pkg.scenes.make_dragon_irregular(int(rng.choice([4000, 30000])), seed=int(rng.randint(1, 1 << 30)))
        elif kind == 2: sd = pkg.scenes.make_blob(int(rng.choice([300, 3000, 9000])), seed=int(rng.randint(1, 1 << 30)))
        else: sd = pkg.scenes.SceneData.load(os.path.join(ROOT, "tests/golden/scenes", str(rng.choice(["cornell", "monkey", "cube"])) + ".npz"))
        # materials: some mirrors (ks.z > 0.01 spawns a reflection, main.cpp:246), random diffuse / specular / shininess
        nm = int(sd.materials.shape[0])
        mats = rng.uniform(0, 1, (nm, 8)).astype(np.float32)
        mats[:, 6] = rng.choice([1.0, 5.0, 20.0, 80.0], nm)  # shininess
        if rng.randint(0, 2): mats[rng.randint(0, nm), 3:6] = 0.0  # a material that reflects nothing
        nl = int(rng.randint(1, 4))
        lights = np.concatenate([rng.uniform(-1.5, 1.5, (nl, 3)), rng.uniform(0.2, 1.0, (nl, 3))], 1).astype(np.float32)
        sd = pkg.scenes.SceneData(pos_nrm=sd.pos_nrm, tri=sd.tri, tri_mesh=sd.tri_mesh, materials=mats, point_lights=lights)
        W, H = int(rng.choice([40, 96, 160])), int(rng.choice([30, 64, 120]))
        cam = np.asarray(pkg.scenes.default_camera(W, H), np.float32).copy()
        cam[3:6] = rng.uniform(-3.0, 3.0, 3).astype(np.float32)   # Euler rotation
        cam[6] = np.float32(rng.uniform(0.6, 4.0))                 # distance: from inside the model's bounds to far outside
        level = int(rng.choice([0, 1, 2, 2, 3, 4]))
        sc = pkg.Scene(sd)
        pkg.set_kernel_shape(0)  # lane per ray
        rgb, st = sc.render(cam, W, H, lights=lights, max_level=level)
        o = orc.OracleScene(sd)
        ref, nrays = o.render(cam, W, H, lights, max_level=level)
        err = float(np.abs(rgb.astype(np.float64) - ref).max())
        stats["max_err"] = max(stats["max_err"], err)
        bad = False
        if not (err <= 1e-5):
            stats["bad_rgb"] += 1; bad = True
        if st["primary_rays"] + st["shadow_rays"] + st["reflection_rays"] != nrays:
            stats["bad_counts"] += 1; bad = True
        if sc.walk():
            stats["certified_frames"] += 1
            pkg.set_kernel_shape(1)  # quad per ray (walk_quad.h)
            rgbq, stq = sc.render(cam, W, H, lights=lights, max_level=level)
            if rgbq.tobytes() != rgb.tobytes():
                stats["shapes_differ"] += 1; bad = True
            pkg.set_kernel_shape(-1)
            sc.set_walk(False)
            rgb0, st0 = sc.render(cam, W, H, lights=lights, max_level=level)
            if rgb0.tobytes() != rgb.tobytes():
                stats["walks_differ"] += 1; bad = True
        pkg.set_kernel_shape(-1)
        if bad:
            print("MISMATCH: iteration", it - 1, "kind", kind, "tris", sd.ntris, f"{W}x{H} level {level} err {err:.3e}", flush=True)
        stats["frames"] += 1; stats["rays"] += int(nrays)
        o.close(); sc.close()
        if verbose and time.time() - t_last > 45:
            t_last = time.time()
            print("progress:", stats, flush=True)
    return stats


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    stats = run(budget, seed0)
    print("render fuzz:", stats, "seed", seed0, "seconds", budget)
    sys.exit(1 if (stats["bad_rgb"] or stats["bad_counts"] or stats["walks_differ"] or stats["shapes_differ"]) else 0)
