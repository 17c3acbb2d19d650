"""GPU parity tests of the CERTIFIED walk (walk_fast.h): fast tree + certificate, exact walk as fallback.  Its results
must be the exact walk's -- hence the oracle's -- bit for bit on every ray: the certificate decides only which of the two
walks produces them.  Covered: scenes whose reference tree ends in single triangles (fast tree forced: zero-thickness
boxes, the F4 false misses, certificates that fail all the time), fat-leaf scenes (the default), equal-t ties and
origin-on-plane acceptances, irregular rays (outside the envelopes: exact walk without trying), finite ray.t, any-hit
soft-shadow rays, and the counters that say how often the fallback ran."""
import os

import numpy as np
import pytest

import rayfam
from conftest import GOLDEN, bits, same_bits
from test_parity_gpu import _assert_hits_equal, _rays

pytestmark = pytest.mark.gpu
FMAX = rayfam.FMAX


@pytest.fixture()
def forced_fast_tree(pkg):
    pkg.set_fast_tree(1)
    yield
    pkg.set_fast_tree(-1)


@pytest.mark.parametrize("name", ["triangle", "cube", "cornell", "monkey", "dodge", "blob"])
def test_forced_fast_tree_on_thin_leaf_scenes(pkg, orc, scene_data, forced_fast_tree, name):
    sd = scene_data(name)
    o = orc.OracleScene(sd)
    _, boxes = o.nodes()
    W = H = 96
    cam = pkg.scenes.default_camera(W, H)
    fam = rayfam.families(sd, boxes, orc.generate_rays(cam, W, H), rng=np.random.RandomState(1234), n_random=3000)
    sc = pkg.Scene(sd)
    assert sc.walk() == 1
    sc.check_layout()
    for k in sorted(fam):
        ref = o.intersect(fam[k])
        hits, normals = sc.intersect(_rays(pkg, fam[k]))
        _assert_hits_equal(hits, normals, ref, f"{name}/{k} certified")
    # the same scene object with the exact walk only: identical bytes
    rays = rayfam.concat(fam)
    h1, n1 = sc.intersect(_rays(pkg, rays))
    sc.set_walk(False)
    assert sc.walk() == 0
    h0, n0 = sc.intersect(_rays(pkg, rays))
    assert h0.tobytes() == h1.tobytes() and n0.tobytes() == n1.tobytes()


def test_f4_false_misses_with_the_fast_tree(pkg, orc, scene_data, forced_fast_tree):
    """The three cube.obj rays on which the reference's BVH misses what its brute force finds.  They
    graze a cube edge and already fail the reference's ROOT gate (the slab test of the root box, bvh.cpp:835-836), which
    both walks share: no tree is walked at all.  The cube's faces lie in zero-thickness leaf boxes: a hit there has
    cur == t exactly on its leaf's box, which the certificate admits (cur <= t, the minimum being unique); what still
    falls back are the hits on the shared diagonal of a face's two triangles (equal t: a tie)."""
    sd = scene_data("cube")
    sc = pkg.Scene(sd)
    assert sc.walk() == 1
    rays = pkg.as_rays(np.broadcast_to(np.float32(rayfam.F4_ORIGIN), (3, 3)), np.float32(rayfam.F4_DIRS))
    hits, _ = sc.intersect(rays)
    assert hits["hit"].tolist() == [0, 0, 0] and hits["prim_id"].tolist() == [pkg.NO_PRIM] * 3
    assert bits(hits["t"]).tolist() == bits(np.float32([FMAX] * 3)).tolist()
    assert sc.count_batch(rays)["tree_rays"] == 0
    W = H = 128
    cam = pkg.scenes.default_camera(W, H)
    prim = orc.generate_rays(cam, W, H)
    c = sc.count_batch(_rays(pkg, prim))
    ref = orc.OracleScene(sd).intersect(prim)
    assert c["tree_rays"] > 0 and c["cert_boxes"] > 0
    assert c["fallback_rays"] < 0.2 * (ref["hit"] == 1).sum()  # hits on axis-aligned faces ARE certified (cur == t)


@pytest.mark.parametrize("ntris", [20_000, 87_000])
def test_dragon_certified_equals_exact_and_oracle(pkg, orc, ntris):
    sd = pkg.scenes.make_dragon(ntris)
    o = orc.OracleScene(sd)
    _, boxes = o.nodes()
    W = H = 128
    cam = pkg.scenes.default_camera(W, H)
    fam = rayfam.families(sd, boxes, orc.generate_rays(cam, W, H), rng=np.random.RandomState(77), n_random=4000)
    sc = pkg.Scene(sd)
    assert sc.walk() == 1, "a fat-leaf scene gets the fast tree by default"
    sc.check_layout()
    for k in sorted(fam):
        hits, normals = sc.intersect(_rays(pkg, fam[k]))
        _assert_hits_equal(hits, normals, o.intersect(fam[k]), f"dragon{ntris}/{k} certified")
    rays = rayfam.concat(fam)
    c1 = sc.count_batch(_rays(pkg, rays))
    h1, n1 = sc.intersect(_rays(pkg, rays))
    sc.set_walk(False)
    c0 = sc.count_batch(_rays(pkg, rays))
    h0, n0 = sc.intersect(_rays(pkg, rays))
    assert h0.tobytes() == h1.tobytes() and n0.tobytes() == n1.tobytes()
    assert c0["fallback_rays"] == 0 and c0["cert_boxes"] == 0 and c0["tree_rays"] == c1["tree_rays"] > 0
    # most rays are certified: the exact walk's inner steps almost disappear
    assert c1["fallback_rays"] < 0.2 * c1["tree_rays"], c1
    assert c1["inner_visits"] < 0.6 * c0["inner_visits"], (c0, c1)  # irregular families (axis-parallel, huge) never try the certified walk
    assert c1["cert_boxes"] > 0


def test_certified_primary_frame_and_tiles(pkg, orc):
    sd = pkg.scenes.make_dragon(60_000)
    W, H = 500, 301
    cam = pkg.scenes.default_camera(W, H)
    sc = pkg.Scene(sd)
    assert sc.walk() == 1
    rays = sc.generate_rays(cam, W, H)
    ref = orc.OracleScene(sd).intersect(rays)
    hits, normals = sc.trace_primary(cam, W, H, want_normals=True)
    _assert_hits_equal(hits, normals, ref, "certified primary frame")
    c = sc.count_primary(cam, W, H)
    assert c["rays"] == W * H and 0 < c["tree_rays"] < W * H and c["fallback_rays"] < 0.05 * c["tree_rays"]
    merged = np.zeros_like(hits)
    for r in range(3):
        part, _ = sc.trace_primary(cam, W, H, rank=r, nranks=3)
        own = pkg.tiling.owned_mask(W, H, r, 3).reshape(-1)
        merged[own] = part[own]
    assert merged.tobytes() == hits.tobytes()
    sc.set_walk(False)
    h0, n0 = sc.trace_primary(cam, W, H, want_normals=True)
    assert h0.tobytes() == hits.tobytes() and n0.tobytes() == normals.tobytes()


def test_certified_ties_and_on_plane(pkg, orc, forced_fast_tree):
    """Coplanar, overlapping, duplicated triangles: equal-t ties and origin-on-plane acceptances are exactly what the
    certificate refuses, so these rays must come out of the exact walk with the reference's tie rules."""
    rng = np.random.RandomState(21)
    n = 40000
    planes_z = np.float32([0.0, 0.25, 0.5])
    rows, tris = [], []
    base_tris = rng.uniform(-1, 1, (40, 3, 2)).astype(np.float32)
    for i in range(n):
        bt = base_tris[rng.randint(0, len(base_tris))]
        z = planes_z[rng.randint(0, 3)]
        for k in range(3):
            rows.append([bt[k, 0], bt[k, 1], z, 0, 0, 1])
        tris.append((3 * i, 3 * i + 1, 3 * i + 2))
    sd = pkg.scenes.SceneData(pos_nrm=np.float32(rows), tri=np.uint32(tris), tri_mesh=np.zeros(n, np.uint32),
                              materials=np.ones((1, 8), np.float32))
    o = orc.OracleScene(sd)
    m = 20000
    oo = np.concatenate([rng.uniform(-1, 1, (m, 2)), rng.choice([-1.0, 0.0, 0.25, 0.5, 0.125, 2.0], (m, 1))], 1).astype(np.float32)
    dd = np.zeros((m, 3), np.float32)
    dd[:, 2] = rng.choice([1.0, -1.0], m)
    dd[:, 0:2] = rng.uniform(-0.3, 0.3, (m, 2))  # no zero components: the rays stay inside the certified envelope
    rays = np.zeros((m, 7), np.float32)
    rays[:, 0:3], rays[:, 3:6], rays[:, 6] = oo, dd, FMAX
    rays[::9, 6] = rng.choice([0.0, 0.1, 0.25, 1.0, -3.0], len(rays[::9]))
    ref = o.intersect(rays)
    sc = pkg.Scene(sd)
    assert sc.walk() == 1
    hits, normals = sc.intersect(_rays(pkg, rays))
    _assert_hits_equal(hits, normals, ref, "certified ties/on-plane")
    c = sc.count_batch(_rays(pkg, rays))
    assert c["fallback_rays"] > 1000  # ties and on-plane origins were refused
    assert (ref["t"][ref["hit"] == 1] == 0).sum() > 500


@pytest.mark.parametrize("seed", [1, 3])
def test_certified_random_multi_mesh_scenes(pkg, orc, forced_fast_tree, seed):
    rng = np.random.RandomState(seed)
    nmesh = rng.randint(5, 40)
    rows, tris, tm = [], [], []
    for m in range(nmesh):
        nt = int(rng.choice([1, 1, 2, 3, 7, 40, 300, 1500]))
        c = rng.uniform(-0.8, 0.8, 3)
        sz = rng.uniform(0.02, 0.5)
        for _ in range(nt):
            p0 = c + rng.uniform(-sz, sz, 3)
            tri = np.stack([p0, p0 + rng.uniform(-0.1, 0.1, 3), p0 + rng.uniform(-0.1, 0.1, 3)])
            kind = rng.randint(0, 40)
            if kind == 0:
                tri[2] = tri[1]
            elif kind == 1 and tris:
                tri = np.asarray(rows[-3:])[:, 0:3]
            base = len(rows)
            nrm = rng.normal(size=(3, 3))
            for k in range(3):
                rows.append(np.concatenate([tri[k], nrm[k] / np.linalg.norm(nrm[k])]))
            tris.append((base, base + 1, base + 2))
            tm.append(m)
    sd = pkg.scenes.SceneData(pos_nrm=np.asarray(rows, np.float32), tri=np.asarray(tris, np.uint32), tri_mesh=np.asarray(tm, np.uint32),
                              materials=rng.uniform(0, 1, (nmesh, 8)).astype(np.float32))
    o, sc = orc.OracleScene(sd), pkg.Scene(sd)
    assert sc.walk() == 1
    sc.check_layout()
    _, b1 = o.nodes()
    W = H = 80
    fam = rayfam.families(sd, b1, orc.generate_rays(pkg.scenes.default_camera(W, H), W, H), rng=rng, n_random=2500)
    rays = rayfam.concat(fam)
    hits, normals = sc.intersect(_rays(pkg, rays))
    _assert_hits_equal(hits, normals, o.intersect(rays), f"certified random scene {seed}")


def test_non_finite_geometry_gets_no_fast_tree(pkg, forced_fast_tree):
    sd = pkg.scenes.make_blob(2000, seed=7)
    pn = sd.pos_nrm.copy()
    pn[5, 1] = np.nan
    sd2 = pkg.scenes.SceneData(pos_nrm=pn, tri=sd.tri, tri_mesh=sd.tri_mesh, materials=sd.materials)
    sc = pkg.Scene(sd2)
    assert sc.walk() == 0
    with pytest.raises(pkg.CgrtError):
        sc.set_walk(True)


def test_render_with_certified_walk(pkg, orc):
    """cgrt_render on a fat-leaf scene: primary, shadow and mirror batches all take the certified walk; RGB within 1e-5 of
    the oracle's recursive driver and byte-identical to the exact-walk render."""
    sd = pkg.scenes.make_dragon(30_000)
    W, H = 160, 120
    cam = pkg.scenes.default_camera(W, H)
    sc = pkg.Scene(sd)
    assert sc.walk() == 1
    rgb1, st1 = sc.render(cam, W, H, max_level=3)
    ref, nrays = orc.OracleScene(sd).render(cam, W, H, sd.point_lights, max_level=3)
    assert np.abs(rgb1.astype(np.float64) - ref).max() <= 1e-5  # tolerance of BASELINE.json's north_star: 1e-5 abs on RGB
    assert st1["primary_rays"] + st1["shadow_rays"] + st1["reflection_rays"] == nrays
    sc.set_walk(False)
    rgb0, st0 = sc.render(cam, W, H, max_level=3)
    assert rgb0.tobytes() == rgb1.tobytes()
    assert (st0["shadow_rays"], st0["reflection_rays"]) == (st1["shadow_rays"], st1["reflection_rays"])
    assert st1["shadow_rays"] > 0


def test_counted_render_reports_the_secondary_work(pkg):
    """cgrt_render_counted: same pixels as cgrt_render; the primary block equals cgrt_count_primary's, the shadow and mirror
    blocks count exactly the rays the frame's statistics report."""
    sd = pkg.scenes.make_dragon(30_000)
    W, H = 160, 120
    cam = pkg.scenes.default_camera(W, H)
    sc = pkg.Scene(sd)
    rgb, st = sc.render(cam, W, H, max_level=3)
    rgb2, st2, work = sc.render_counted(cam, W, H, max_level=3)
    assert rgb2.tobytes() == rgb.tobytes()
    assert work["primary"]["rays"] == W * H == st2["primary_rays"]
    assert work["shadow"]["rays"] == st2["shadow_rays"] > 0 and work["mirror"]["rays"] == st2["reflection_rays"] > 0
    cp = sc.count_primary(cam, W, H)
    for k in ("tree_rays", "sub_visits", "tri_tests", "cert_boxes", "fallback_rays"):
        assert work["primary"][k] == cp[k]
    assert work["shadow"]["tree_rays"] > 0 and work["mirror"]["sub_visits"] > 0


def test_axis_parallel_rays_are_certified_too(pkg, orc):
    """Rays with zero direction components lie outside the exact fast division's envelope (RayFast::fd); their certificates
    use the reference's box test as written (IEEE divisions, NaN-order-sensitive ternaries) and they still take the
    certified walk: an orthographic grid straight down the z axis onto the dragon, bit-exact, with (almost) no exact steps."""
    sd = pkg.scenes.make_dragon(60_000)
    n = 200
    xs, ys = np.meshgrid(np.linspace(-0.9, 0.9, n, dtype=np.float32), np.linspace(-0.9, 0.9, n, dtype=np.float32))
    rays = np.zeros((n * n, 7), np.float32)
    rays[:, 0], rays[:, 1], rays[:, 2] = xs.ravel(), ys.ravel(), -3.0
    rays[:, 5] = 1.0  # direction (0, 0, 1): two zero components -> +-inf / NaN slab parameters upstream
    rays[1::2, 5] = 0.5
    rays[:, 6] = FMAX
    sc, o = pkg.Scene(sd), orc.OracleScene(sd)
    hits, normals = sc.intersect(_rays(pkg, rays))
    _assert_hits_equal(hits, normals, o.intersect(rays), "orthographic grid")
    c = sc.count_batch(_rays(pkg, rays))
    assert c["tree_rays"] > 0.3 * n * n and (hits["hit"] == 1).sum() > 0.1 * n * n
    assert c["fallback_rays"] < 0.02 * c["tree_rays"], c
    sc.set_walk(False)
    c0 = sc.count_batch(_rays(pkg, rays))
    assert c["inner_visits"] < 0.05 * c0["inner_visits"]


def test_irregular_standin_frame(pkg, orc):
    """The scan-like variant of the stand-in (irregular density, slivers, noise, shuffled order): a 480x270 frame against the
    oracle, certified walk by default, few fallbacks."""
    sd = pkg.scenes.make_dragon_irregular(120_000)
    W, H = 480, 270
    cam = pkg.scenes.default_camera(W, H)
    sc = pkg.Scene(sd)
    assert sc.walk() == 1
    hits, normals = sc.trace_primary(cam, W, H, want_normals=True)
    _assert_hits_equal(hits, normals, orc.OracleScene(sd).intersect(sc.generate_rays(cam, W, H)), "irregular stand-in")
    c = sc.count_primary(cam, W, H)
    assert c["fallback_rays"] < 0.01 * c["tree_rays"] and (hits["hit"] == 1).mean() > 0.05
