"""Parity fuzz (run on the GPU box): random scenes x ray families, device (default build: accelerators + certified walk, then the
exact walk on the same scene) against the oracle, bit for bit (flag, t bits, primitive id, material, normal bits).  Scenes: random
multi-mesh soups with degenerate and duplicated triangles, sliver scenes, dragon stand-ins of random size (regular and
irregular), each at a random power-of-two scale; rays: tests/rayfam.py families + rays aimed at vertices / edges with ulp
perturbations (tests/test_adversarial_gpu.py).  Since round 3 every certified scene is also run in the quad-per-ray kernel shape.
Usage: python tools/fuzz_parity.py [seconds] [seed]; tests/test_fuzz_gpu.py calls run() over the committed seed list."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as e
pkg = e.load_package(); orc = e.load_oracle()
import rayfam
import test_adversarial_gpu as adv


def soup(rng):
    nmesh = rng.randint(1, 30)
    rows, tris, tm = [], [], []
    for m in range(nmesh):
        nt = int(rng.choice([1, 2, 5, 30, 200, 1500, 6000]))
        c = rng.uniform(-0.8, 0.8, 3); sz = rng.uniform(0.02, 0.5)
        for _ in range(nt):
            p0 = c + rng.uniform(-sz, sz, 3)
            tri = np.stack([p0, p0 + rng.uniform(-0.1, 0.1, 3), p0 + rng.uniform(-0.1, 0.1, 3)])
            kind = rng.randint(0, 40)
            if kind == 0: tri[2] = tri[1]
            elif kind == 1 and tris: tri = np.asarray(rows[-3:])[:, 0:3]
            elif kind == 2: tri[:, rng.randint(0, 3)] = tri[0, 0]  # axis-aligned
            base = len(rows); nrm = rng.normal(size=(3, 3))
            for k in range(3): rows.append(np.concatenate([tri[k], nrm[k] / np.linalg.norm(nrm[k])]))
            tris.append((base, base + 1, base + 2)); tm.append(m)
    return pkg.scenes.SceneData(pos_nrm=np.asarray(rows, np.float32), tri=np.asarray(tris, np.uint32), tri_mesh=np.asarray(tm, np.uint32),
                                materials=rng.uniform(0, 1, (nmesh, 8)).astype(np.float32))


def same(h, n, ref):
    ok = np.array_equal(h["hit"], ref["hit"]) and np.array_equal(h["prim_id"], ref["prim"]) and np.array_equal(h["material_id"], ref["material"])
    a, b = h["t"], ref["t"]
    ok = ok and bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())
    m = ref["hit"] == 1
    x, y = n[m], ref["normal"][m]
    return ok and bool(((x.view(np.uint32) == y.view(np.uint32)) | (np.isnan(x) & np.isnan(y))).all())



def run(budget=120.0, seed0=1, verbose=True):
    """Fuzz for `budget` seconds from seed `seed0`; returns the statistics (mismatching_batches must be 0)."""
    t_end = time.time() + budget
    t_last = time.time()
    stats = dict(scenes=0, rays=0, certified_scenes=0, fallback_rays=0, tree_rays=0, quad_shape_batches=0, hinted_frames=0, mismatching_batches=0)
    it = 0
    while time.time() < t_end:
        rng = np.random.RandomState(seed0 * 100003 + it); it += 1
        kind = rng.randint(0, 5)
        if kind == 0: sd = soup(rng)
        elif kind == 1: sd = adv._sliver_scene(pkg, rng, ntris=int(rng.choice([300, 1500, 5000])), with_dragon=bool(rng.randint(0, 2)))
        elif kind == 2: sd = pkg.scenes.make_dragon(int(rng.choice([3000, 20000, 90000])), seed=int(rng.randint(1, 1 << 30)))
        elif kind == 3: sd = pkg.scenes.make_dragon_irregular(int(rng.choice([5000, 40000])), seed=int(rng.randint(1, 1 << 30)))
        else: sd = pkg.scenes.make_blob(int(rng.choice([200, 2000, 9000])), seed=int(rng.randint(1, 1 << 30)))
        k = int(rng.choice([0, 0, 0, -20, -8, 7, 19, 30, 37]))
        sc_ = np.float32(2.0) ** np.float32(k)
        pn = sd.pos_nrm.copy(); pn[:, 0:3] = (pn[:, 0:3] * sc_).astype(np.float32)
        sd = pkg.scenes.SceneData(pos_nrm=pn, tri=sd.tri, tri_mesh=sd.tri_mesh, materials=sd.materials)
        o = orc.OracleScene(sd); _, boxes = o.nodes()
        W = H = int(rng.choice([32, 64, 96]))
        cam = np.asarray(pkg.scenes.default_camera(W, H), np.float32).copy(); cam[6] *= float(sc_)  # trackball distance scales with the scene
        fam = rayfam.families(sd, boxes[np.isfinite(boxes).all(1)], orc.generate_rays(cam, W, H), rng=rng, n_random=800)
        rays = rayfam.concat(fam)
        aimed = adv._aimed_rays(sd, rng, n=300)
        rays = np.concatenate([rays, aimed]).astype(np.float32)
        rays = rays[np.isfinite(rays[:, 0:6]).all(1)]
        ref = o.intersect(rays)
        sc = pkg.Scene(sd)
        R = rays.view(pkg.RAY_DTYPE).reshape(-1)
        pkg.set_kernel_shape(0)  # lane per ray
        h, n = sc.intersect(R)
        bad = not same(h, n, ref)
        if sc.walk():
            c = sc.count_batch(R); stats["certified_scenes"] += 1; stats["fallback_rays"] += c["fallback_rays"]; stats["tree_rays"] += c["tree_rays"]
            pkg.set_kernel_shape(1)  # quad per ray (walk_quad.h): the same bytes
            hq, nq = sc.intersect(R)
            bad = bad or not same(hq, nq, ref)
            stats["quad_shape_batches"] += 1
            pkg.set_kernel_shape(-1)
            # frame hints (cgrt_set_frame_hints): frames of one shape one after the other, hard tiles first / as 16-ray waves, with
            # thresholds low enough that tiles are listed in frames this small -- the same bytes as the plain frame
            Wf, Hf = int(rng.choice([64, 200, 333])), int(rng.choice([48, 120, 211]))
            camf = np.asarray(pkg.scenes.default_camera(Wf, Hf), np.float32).copy(); camf[6] *= float(sc_)
            camf[3:6] = rng.uniform(-3.0, 3.0, 3).astype(np.float32)
            pkg.set_frame_hints(0)
            f0 = sc.trace_primary(camf, Wf, Hf, want_normals=True)
            pkg.debug_set_hint_thresholds(int(rng.choice([20, 100, 400])), int(rng.choice([20, 60, 200])))
            for mode in (1, 2):
                pkg.set_frame_hints(mode)
                for _ in range(7):  # (a shape gets its hint buffers at the third frame in a row)
                    f = sc.trace_primary(camf, Wf, Hf, want_normals=True)
                    bad = bad or f[0].tobytes() != f0[0].tobytes() or f[1].tobytes() != f0[1].tobytes()
                    stats["hinted_frames"] += 1
            pkg.set_frame_hints(-1)
            pkg.debug_set_hint_thresholds(0, 0)
            sc.set_walk(False)
            h0, n0 = sc.intersect(R)
            bad = bad or not same(h0, n0, ref)
        pkg.set_kernel_shape(-1)
        hb, nb = sc.intersect_brute(R[:2000]) if sd.ntris <= 20000 else (None, None)
        if hb is not None:
            bad = bad or not same(hb, nb, o.intersect(rays[:2000], brute_force=True))
        stats["scenes"] += 1; stats["rays"] += len(rays)
        if bad:
            stats["mismatching_batches"] += 1
            print("MISMATCH: iteration", it - 1, "kind", kind, "scale 2^%d" % k, "tris", sd.ntris, flush=True)
        o.close(); sc.close()
        if verbose and time.time() - t_last > 45:
            t_last = time.time()
            print("progress:", stats, flush=True)
    return stats


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    stats = run(budget, seed0)
    print("fuzz:", stats, "seed", seed0, "seconds", budget)
    sys.exit(1 if stats["mismatching_batches"] else 0)
