"""GPU parity tests proper (-m gpu): the HIP path, called through the C-ABI, against the CPU oracle on the
same seeded inputs and against the committed golden fixtures.  Bar: bit-exact hit flag, t (u32 pattern),
primitive id, material id; hitInfo.normal bit-exact too (it is computed with the reference's double-precision
area formula on both sides)."""
import os

import numpy as np
import pytest

import rayfam
from conftest import GOLDEN, bits, same_bits

pytestmark = pytest.mark.gpu
FMAX = rayfam.FMAX


def _rays(pkg, r7):
    return np.ascontiguousarray(r7, np.float32).view(pkg.RAY_DTYPE).reshape(-1)


def _assert_hits_equal(hits, normals, ref, what=""):
    assert np.array_equal(hits["hit"], ref["hit"]), f"{what}: hit flags differ on {(hits['hit'] != ref['hit']).sum()} rays"
    bad = ~same_bits(hits["t"], ref["t"])
    assert not bad.any(), f"{what}: t bits differ on {bad.sum()} rays, first {np.nonzero(bad)[0][:5]}"
    assert np.array_equal(hits["prim_id"], ref["prim"]), f"{what}: primitive ids differ"
    assert np.array_equal(hits["material_id"], ref["material"]), f"{what}: material ids differ"
    if normals is not None:
        m = ref["hit"] == 1
        assert same_bits(normals[m], ref["normal"][m]).all(), f"{what}: normals differ"


# ---------------------------------------------------------------------------------------------------
# primitives (src/ray_tracing.h:10-20)
# ---------------------------------------------------------------------------------------------------
def _random_rays(rng, n, scale=1.0):
    r = np.zeros((n, 7), np.float32)
    r[:, 0:3] = rng.uniform(-2, 2, (n, 3)) * scale
    d = rng.normal(size=(n, 3))
    r[:, 3:6] = d / np.linalg.norm(d, axis=1, keepdims=True)
    r[:, 6] = FMAX
    return r


def test_ray_triangle_primitive(pkg, orc):
    rng = np.random.RandomState(11)
    n = 200_000
    tri = rng.uniform(-1, 1, (n, 18)).astype(np.float32)
    r = _random_rays(rng, n)
    # aim most rays at a point of their triangle so that hits are common
    w = rng.dirichlet((1, 1, 1), n).astype(np.float32)
    tgt = w[:, 0:1] * tri[:, 0:3] + w[:, 1:2] * tri[:, 3:6] + w[:, 2:3] * tri[:, 6:9]
    d = tgt - r[:, 0:3]
    r[: n // 2, 3:6] = (d / np.linalg.norm(d, axis=1, keepdims=True))[: n // 2]
    r[::7, 6] = rng.uniform(0, 3, len(r[::7]))  # finite t
    # exact on-plane origins on axis-aligned triangles, scaled directions, denormal-ish geometry
    tri[:1000, [2, 5, 8]] = 0.5
    r[:1000, 2] = 0.5
    tri[1000:2000] *= np.float32(1e-18)
    r[1000:2000, 0:3] *= np.float32(1e-18)
    r[2000:3000, 3:6] *= np.float32(1e-20)
    r[3000:4000, 3:6] *= np.float32(1e15)
    t, hit, nrm = pkg.ray_triangle(tri, _rays(pkg, r))
    ref = orc.ray_triangle(tri, r)
    assert 0.2 < hit.mean() < 0.8
    assert np.array_equal(hit, ref["hit"].astype(np.uint8))
    assert same_bits(t, ref["t"]).all()
    m = hit == 1
    assert same_bits(nrm[m], ref["normal"][m]).all()


def test_ray_plane_and_triangle_plane_and_point_in_triangle(pkg, orc):
    rng = np.random.RandomState(12)
    n = 100_000
    tri9 = rng.uniform(-1, 1, (n, 9)).astype(np.float32)
    tri9[:500] *= np.float32(1e-15)  # tiny triangles: normalisation of tiny cross products
    tri9[500:600, 3:6] = tri9[500:600, 0:3]  # degenerate: NaN normal
    pl = pkg.triangle_plane(tri9)
    assert same_bits(pl, orc.triangle_plane(tri9)).all()
    r = _random_rays(rng, n)
    r[::5, 6] = rng.uniform(0, 3, len(r[::5]))
    r[:2000, 3:6] = 0.0  # zero direction
    pl2 = pl.copy()
    pl2[5000:6000, 0] = (r[5000:6000, 0:3] * pl2[5000:6000, 1:4]).astype(np.float32).sum(1)  # near on-plane
    t, hit = pkg.ray_plane(pl2, _rays(pkg, r))
    ref = orc.ray_plane(pl2, r)
    assert np.array_equal(hit, ref["hit"].astype(np.uint8)) and same_bits(t, ref["t"]).all()
    assert np.isnan(t).sum() > 50  # degenerate planes were exercised
    pin = np.concatenate([tri9, pl[:, 1:4], rng.uniform(-1, 1, (n, 3)).astype(np.float32)], 1)
    assert np.array_equal(pkg.point_in_triangle(pin), orc.point_in_triangle(pin))


def test_ray_box_primitive(pkg, orc):
    rng = np.random.RandomState(13)
    n = 200_000
    lo = rng.uniform(-1, 0.5, (n, 3)).astype(np.float32)
    hi = (lo + rng.uniform(0, 1, (n, 3))).astype(np.float32)
    hi[:5000, 0] = lo[:5000, 0]  # zero-thickness boxes (F4)
    box = np.concatenate([lo, hi], 1)
    r = _random_rays(rng, n)
    # zero / negative-zero direction components, origins on faces, inside, finite t
    r[0:20000:2, 3] = 0.0
    r[1:20000:2, 4] = -0.0
    r[20000:30000, 3:5] = 0.0
    r[30000:40000, 0] = lo[30000:40000, 0]
    r[40000:50000, 0:3] = ((lo[40000:50000].astype(np.float64) + hi[40000:50000]) / 2).astype(np.float32)
    r[50000:60000, 1] = hi[50000:60000, 1]
    r[50000:60000, 4] = 0.0  # origin in the face plane AND parallel to it => 0/0 = NaN
    r[::3, 6] = rng.uniform(0, 3, len(r[::3]))
    t, hit, inside = pkg.ray_box(box, _rays(pkg, r))
    ref = orc.ray_box(box, r)
    assert np.array_equal(hit, ref["hit"].astype(np.uint8))
    assert same_bits(t, ref["t"]).all()
    assert np.array_equal(inside, ref["pad"].astype(np.uint8))
    assert 0.05 < hit.mean() < 0.95 and inside.sum() > 5000


def test_ray_sphere_primitive(pkg, orc):
    rng = np.random.RandomState(14)
    n = 100_000
    sph = np.concatenate([rng.uniform(-1, 1, (n, 3)), rng.uniform(0.05, 1.5, (n, 1))], 1).astype(np.float32)
    r = _random_rays(rng, n)
    d = sph[:, 0:3] - r[:, 0:3]
    r[: n // 2, 3:6] = (d / np.linalg.norm(d, axis=1, keepdims=True))[: n // 2]
    r[::4, 6] = rng.uniform(0, 3, len(r[::4]))
    t, hit, nrm = pkg.ray_sphere(sph, _rays(pkg, r))
    ref = orc.ray_sphere(sph, r)
    assert np.array_equal(hit, ref["hit"].astype(np.uint8)) and same_bits(t, ref["t"]).all()
    m = hit == 1
    assert same_bits(nrm[m], ref["normal"][m]).all()


def test_fast_division_is_ieee(pkg):
    """The kernels' 4-operation exact division (Markstein correction with a per-ray reciprocal) must equal
    IEEE a / d bit for bit over its whole admitted envelope: |d| in [2^-36, 2^36], a = 0 or |a| in [2^-64, 2^41].
    Random mantissas/exponents plus the classic hard cases: denominators with all-ones significands, powers of
    two, numerators one ulp around exact multiples (quotients next to rounding midpoints)."""
    rng = np.random.RandomState(77)
    n = 1 << 23

    def rnd(nn, emin, emax):
        m = rng.randint(0, 1 << 23, nn).astype(np.uint32)
        e = rng.randint(emin + 127, emax + 127 + 1, nn).astype(np.uint32)
        sg = rng.randint(0, 2, nn).astype(np.uint32)
        return ((sg << 31) | (e << 23) | m).view(np.float32)

    d = rnd(n, -36, 35)
    a = rnd(n, -64, 40)
    # hard denominators
    d[: 1 << 18] = ((rng.randint(91, 163, 1 << 18).astype(np.uint32) << 23) | np.uint32(0x7FFFFF)).view(np.float32)  # 1.11..1
    d[1 << 18: 1 << 19] = ((rng.randint(91, 163, 1 << 18).astype(np.uint32) << 23)).view(np.float32)  # powers of two
    d[1 << 19: 1 << 20] = ((rng.randint(91, 163, 1 << 19).astype(np.uint32) << 23) | rng.randint(0, 4, 1 << 19).astype(np.uint32)).view(np.float32)
    # numerators that make the quotient land next to a representable value or a midpoint: a = RN(q*d) +- ulps
    q = rnd(1 << 21, -20, 20)
    prod = (q.astype(np.float64) * d[1 << 21: 1 << 22].astype(np.float64)).astype(np.float32)
    bump = rng.randint(-2, 3, 1 << 21).astype(np.int32)
    a[1 << 21: 1 << 22] = (prod.view(np.int32) + bump).view(np.float32)
    a[1 << 22: (1 << 22) + 4096] = 0.0
    keep = np.isfinite(a) & np.isfinite(d) & ((a == 0) | ((np.abs(a) >= 2.0 ** -64) & (np.abs(a) <= 2.0 ** 41)))
    keep &= (np.abs(d) >= 2.0 ** -36) & (np.abs(d) <= 2.0 ** 36)
    mism, bad = pkg.fastdiv_check(a[keep], d[keep])
    assert keep.sum() > 0.95 * n
    assert mism == 0, f"{mism} mismatches, first: a={bad[0]!r} d={bad[1]!r} got={bad[2]!r} want={bad[3]!r}"


# ---------------------------------------------------------------------------------------------------
# BoundingVolumeHierarchy::intersect, batched
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["triangle", "cube", "cornell", "monkey", "blob", "spheres"])
def test_intersect_matches_golden_fixture(pkg, scene_data, name):
    z = np.load(os.path.join(GOLDEN, f"hits_{name}.npz"))
    sc = pkg.Scene(scene_data(name))
    assert sc.num_levels() == int(z["levels"])
    hits, normals = sc.intersect(_rays(pkg, z["rays"]))
    assert np.array_equal(hits["hit"], z["hit"])
    assert same_bits(hits["t"], z["t_bits"].view(np.float32)).all()
    assert np.array_equal(hits["prim_id"], z["prim"])
    assert np.array_equal(hits["material_id"], z["material"])
    m = z["hit"] == 1
    assert same_bits(normals[m], z["normal_bits"].view(np.float32)[m]).all()


@pytest.mark.parametrize("name", ["cube", "cornell", "monkey", "dodge", "blob"])
def test_intersect_matches_oracle_fresh_rays(pkg, orc, scene_data, name):
    sd = scene_data(name)
    o = orc.OracleScene(sd)
    _, boxes = o.nodes()
    W = H = 96
    cam = pkg.scenes.default_camera(W, H)
    fam = rayfam.families(sd, boxes, orc.generate_rays(cam, W, H), rng=np.random.RandomState(99), n_random=4000)
    sc = pkg.Scene(sd)
    for k in sorted(fam):
        hits, normals = sc.intersect(_rays(pkg, fam[k]))
        _assert_hits_equal(hits, normals, o.intersect(fam[k]), f"{name}/{k}")


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_multi_mesh_scenes_with_degenerates(pkg, orc, seed):
    """Many meshes of very different sizes (multi-mesh nodes and leaves, single-triangle meshes), zero-area and
    duplicated triangles, a NaN and an infinite vertex coordinate: builder and traversal against the oracle."""
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
                tri[2] = tri[1]  # zero area
            elif kind == 1 and tris:
                tri = np.asarray(rows[-3:])[:, 0:3]  # exact duplicate of the previous triangle
            base = len(rows)
            nrm = rng.normal(size=(3, 3))
            for k in range(3):
                rows.append(np.concatenate([tri[k], nrm[k] / np.linalg.norm(nrm[k])]))
            tris.append((base, base + 1, base + 2))
            tm.append(m)
    pn = np.asarray(rows, np.float32)
    if seed == 2:
        pn[5, 1] = np.nan
        pn[len(pn) // 2, 0] = np.inf
    sd = pkg.scenes.SceneData(pos_nrm=pn, tri=np.asarray(tris, np.uint32), tri_mesh=np.asarray(tm, np.uint32),
                              materials=rng.uniform(0, 1, (nmesh, 8)).astype(np.float32))
    o, sc = orc.OracleScene(sd), pkg.Scene(sd)
    m1, b1 = o.nodes()
    m2, b2 = sc.nodes()
    assert np.array_equal(m1, m2) and same_bits(b1, b2).all()
    finite_boxes = b1[np.isfinite(b1).all(1)]
    W = H = 80
    fam = rayfam.families(sd, finite_boxes, orc.generate_rays(pkg.scenes.default_camera(W, H), W, H), rng=rng, n_random=2500)
    rays = rayfam.concat(fam)
    rays = rays[np.isfinite(rays).all(1) | (rays[:, 6] == np.inf)]
    hits, normals = sc.intersect(_rays(pkg, rays))
    _assert_hits_equal(hits, normals, o.intersect(rays), f"random scene {seed}")
    assert hits["hit"].mean() > 0.1


@pytest.mark.parametrize("name", ["monkey", "blob"])
def test_non_finite_rays(pkg, orc, scene_data, name):
    """NaN / +-inf in origin, direction or ray.t: every comparison with a NaN is false on both sides, the walk must
    terminate and agree with the oracle (t compared NaN-aware)."""
    sd = scene_data(name)
    rng = np.random.RandomState(31)
    n = 6000
    rays = np.zeros((n, 7), np.float32)
    rays[:, 0:3] = rng.uniform(-2, 2, (n, 3))
    d = rng.uniform(-0.5, 0.5, (n, 3)) - rays[:, 0:3]
    rays[:, 3:6] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 6] = FMAX
    special = np.float32([np.nan, np.inf, -np.inf])
    for i in range(n):
        k = i % 8
        if k < 7:
            rays[i, k] = special[(i // 8) % 3]
        else:  # two poisoned components
            rays[i, rng.randint(0, 3)] = np.nan
            rays[i, 3 + rng.randint(0, 3)] = np.inf
    o = orc.OracleScene(sd)
    ref = o.intersect(rays)
    hits, normals = pkg.Scene(sd).intersect(_rays(pkg, rays))
    assert np.array_equal(hits["hit"], ref["hit"])
    assert same_bits(hits["t"], ref["t"]).all()
    assert np.array_equal(hits["prim_id"], ref["prim"]) and np.array_equal(hits["material_id"], ref["material"])
    m = ref["hit"] == 1
    assert same_bits(normals[m], ref["normal"][m]).all()


def test_f4_false_misses_reproduced_on_gpu(pkg, scene_data):
    sc = pkg.Scene(scene_data("cube"))
    rays = pkg.as_rays(np.broadcast_to(np.float32(rayfam.F4_ORIGIN), (3, 3)), np.float32(rayfam.F4_DIRS))
    hits, _ = sc.intersect(rays)
    assert hits["hit"].tolist() == [0, 0, 0] and hits["prim_id"].tolist() == [pkg.NO_PRIM] * 3
    assert bits(hits["t"]).tolist() == bits(np.float32([FMAX] * 3)).tolist()


def test_miss_leaves_normals_untouched_and_empty_batch(pkg, scene_data):
    sc = pkg.Scene(scene_data("cube"))
    rays = pkg.as_rays([[5, 5, 5]], [[1, 0, 0]])
    hits = np.zeros(1, pkg.HIT_DTYPE)
    nrm = np.full((1, 3), 7.5, np.float32)
    import ctypes as C

    rc = pkg.lib().cgrt_intersect_batch(sc._h, rays.ctypes.data_as(C.c_void_p), 1, hits.ctypes.data_as(C.c_void_p),
                                        nrm.ctypes.data_as(C.c_void_p))
    assert rc == 0 and hits["hit"][0] == 0 and nrm.tolist() == [[7.5, 7.5, 7.5]]  # HitInfo untouched on a miss
    h, _ = sc.intersect(np.zeros(0, pkg.RAY_DTYPE))
    assert len(h) == 0


def test_mixed_meshes_and_spheres(pkg, orc, scene_data):
    sd = scene_data("cornell")
    sd2 = pkg.scenes.SceneData(pos_nrm=sd.pos_nrm, tri=sd.tri, tri_mesh=sd.tri_mesh, materials=sd.materials,
                               spheres=np.float32([[0.1, -0.2, 0.0, 0.25, -1], [-0.3, 0.2, 0.1, 0.2, -1]]))
    o, sc = orc.OracleScene(sd2), pkg.Scene(sd2)
    W = H = 160
    r = orc.generate_rays(pkg.scenes.default_camera(W, H), W, H)
    hits, normals = sc.intersect(_rays(pkg, r))
    ref = o.intersect(r)
    _assert_hits_equal(hits, normals, ref, "cornell+spheres")
    assert (hits["prim_id"][hits["hit"] == 1] >= sd.ntris).sum() > 100  # spheres were hit
    # a sphere-over-triangle hit keeps the triangle's material (bvh.cpp:878-879 never writes it)
    assert ((hits["prim_id"] >= sd.ntris) & (hits["prim_id"] != pkg.NO_PRIM) & (hits["material_id"] >= 0)).sum() > 0


# ---------------------------------------------------------------------------------------------------
# fused primary frame (renderRayTracing's ray generation + intersect)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,W,H", [("cube", 256, 256), ("monkey", 1024, 1024), ("cornell", 480, 270), ("blob", 333, 217)])
def test_trace_primary_bit_exact(pkg, orc, scene_data, name, W, H):
    """Config 1 (cube 256^2) and config 2 (monkey 1024^2): bit-exact flag / t / id vs the CPU path."""
    sd = scene_data(name)
    cam = pkg.scenes.default_camera(W, H)
    sc, o = pkg.Scene(sd), orc.OracleScene(sd)
    rays = sc.generate_rays(cam, W, H)
    assert np.array_equal(rays.view(np.float32).view(np.uint32).reshape(-1, 7), bits(orc.generate_rays(cam, W, H)))
    hits, normals = sc.trace_primary(cam, W, H, want_normals=True)
    _assert_hits_equal(hits, normals, o.intersect(rays), f"{name} {W}x{H}")
    assert hits["hit"].sum() > 0.03 * W * H


def test_trace_primary_tiles_partition_the_image(pkg, scene_data):
    """Image tiling across ranks: ranks own disjoint 64x64 super-tiles whose union is the
    frame, and the device's ownership equals the host-side tiling.owned_mask bench.py relies on."""
    from cg_raytracer_amd import tiling

    sd = scene_data("monkey")
    W, H = 500, 301  # ragged right/bottom tiles
    cam = pkg.scenes.default_camera(W, H)
    sc = pkg.Scene(sd)
    full, _ = sc.trace_primary(cam, W, H)
    for nranks in (2, 3, 8):
        owner = np.full(W * H, -1)
        merged = np.zeros(W * H, pkg.HIT_DTYPE)
        for rank in range(nranks):
            part, _ = sc.trace_primary(cam, W, H, rank=rank, nranks=nranks)
            mine = ~np.isnan(part["t"])
            assert np.array_equal(mine.reshape(H, W), tiling.owned_mask(W, H, rank, nranks))
            assert np.all(owner[mine] == -1)
            owner[mine] = rank
            merged[mine] = part[mine]
        assert np.all(owner >= 0)
        assert merged.tobytes() == full.tobytes()
        assert np.bincount(owner).min() > 0.4 * W * H / nranks
    # sub-rectangle
    part, _ = sc.trace_primary(cam, W, H, rect=(40, 16, 211, 100))
    yy, xx = np.divmod(np.arange(W * H), W)
    inside = (xx >= 40) & (xx < 211) & (yy >= 16) & (yy < 100)
    assert np.array_equal(~np.isnan(part["t"]), inside)
    assert part[inside].tobytes() == full[inside].tobytes()


@pytest.mark.parametrize("W,H,nranks", [(640, 360, 1), (333, 217, 1), (500, 301, 3), (64, 64, 1), (1, 1, 1)])
def test_persistent_kernel_same_results(pkg, orc, W, H, nranks):
    """Primary mode 1 (persistent waves pulling tiles from per-XCD queues, finished lanes refilled) must write
    exactly what mode 0 (one wave per tile) writes -- scheduling is a speed matter only -- and what the oracle says."""
    sd = pkg.scenes.make_dragon(40_000)
    cam = pkg.scenes.default_camera(W, H)
    sc = pkg.Scene(sd)
    sc.set_walk(False)  # the persistent variant walks exactly; its step counts are compared with the exact walk's
    ref = orc.OracleScene(sd).intersect(sc.generate_rays(cam, W, H))
    try:
        for rank in range(nranks):
            pkg.set_primary_mode(0)
            h0, n0 = sc.trace_primary(cam, W, H, rank=rank, nranks=nranks, want_normals=True)
            c0 = sc.count_primary(cam, W, H, rank=rank, nranks=nranks)
            pkg.set_primary_mode(1)
            for rep in range(3):  # queue blocks are reused round-robin and must come back clean
                h1, n1 = sc.trace_primary(cam, W, H, rank=rank, nranks=nranks, want_normals=True)
                assert h1.tobytes() == h0.tobytes() and n1.tobytes() == n0.tobytes()
            c1 = sc.count_primary(cam, W, H, rank=rank, nranks=nranks)
            assert c1["rays"] == c0["rays"] and c1["inner_visits"] == c0["inner_visits"] and c1["leaf_visits"] == c0["leaf_visits"]
            own = ~np.isnan(h1["t"])
            _assert_hits_equal(h1[own], n1[own], ref[own], f"persistent {W}x{H} rank {rank}/{nranks}")
    finally:
        pkg.set_primary_mode(0)


def test_persistent_kernel_many_frames_in_flight(pkg):
    """More launches in flight than the persistent kernel has queue blocks (8): twelve host threads trace the same frame
    at once, each on its call lane's private stream.  A block is reused only behind the launch that had it last, so
    every frame must come out like the one-wave-per-tile frame."""
    import threading

    W, H = 320, 200
    sd = pkg.scenes.make_dragon(40_000)
    cam = pkg.scenes.default_camera(W, H)
    sc = pkg.Scene(sd)
    h0, _ = sc.trace_primary(cam, W, H)
    bad = []

    def work():
        for _ in range(4):
            h1, _ = sc.trace_primary(cam, W, H)
            if h1.tobytes() != h0.tobytes():
                bad.append(1)

    try:
        pkg.set_primary_mode(1)
        threads = [threading.Thread(target=work) for _ in range(12)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        pkg.set_primary_mode(0)
    assert not bad, f"{len(bad)} of 48 concurrent persistent frames differ"


def test_counters_match_oracle_replay(pkg, orc):
    """cgrt_count_* (instrumented launch) == the oracle's count of the same traversal: the reference tree is
    walked node for node like the reference does; with the in-leaf accelerator off the triangle tests match
    too, with it on they can only go down."""
    sd = pkg.scenes.make_dragon(20_000)
    W = H = 200
    cam = pkg.scenes.default_camera(W, H)
    o = orc.OracleScene(sd)
    _, cnt = o.intersect(orc.generate_rays(cam, W, H), counters=True)
    try:
        pkg.set_leaf_accel(False)
        lin = pkg.Scene(sd)
    finally:
        pkg.set_leaf_accel(True)
    acc = pkg.Scene(sd)
    acc.set_walk(False)  # the exact walk is the one that takes the reference's steps (the certified walk has none to count)
    assert lin.num_subnodes() == 0 and acc.num_subnodes() > 0 and lin.walk() == 0
    got = lin.count_primary(cam, W, H)
    assert got["rays"] == W * H and got["sub_visits"] == 0
    assert got["inner_visits"] == cnt["inner_visits"] and got["leaf_visits"] == cnt["leaf_visits"]
    assert got["tri_tests"] == cnt["tri_tests"]
    got2 = lin.count_batch(lin.generate_rays(cam, W, H))
    assert {k: got2[k] for k in ("inner_visits", "leaf_visits", "tri_tests")} == {k: got[k] for k in ("inner_visits", "leaf_visits", "tri_tests")}
    ga = acc.count_primary(cam, W, H)
    assert ga["inner_visits"] == cnt["inner_visits"] and ga["leaf_visits"] == cnt["leaf_visits"]
    assert 0 < ga["tri_tests"] < 0.5 * cnt["tri_tests"] and ga["sub_visits"] > 0


@pytest.mark.parametrize("sub_leaf", [1, 4, 16])
def test_leaf_accelerator_does_not_change_results(pkg, orc, sub_leaf):
    """Same bits with the in-leaf accelerator on (any sub-leaf size) and off, on every ray family -- incl.
    irregular rays (far origins, tiny/huge directions) that take the test-everything path inside leaves."""
    sd = pkg.scenes.make_dragon(30_000)
    o = orc.OracleScene(sd)
    _, boxes = o.nodes()
    W = H = 64
    fam = rayfam.families(sd, boxes, orc.generate_rays(pkg.scenes.default_camera(W, H), W, H), rng=np.random.RandomState(5),
                          n_random=3000)
    rays = rayfam.concat(fam)
    ref = o.intersect(rays)
    try:
        pkg.set_leaf_accel(False)
        lin = pkg.Scene(sd)
        pkg.set_leaf_accel(True, sub_leaf)
        acc = pkg.Scene(sd)
    finally:
        pkg.set_leaf_accel(True)
    h0, n0 = lin.intersect(_rays(pkg, rays))
    h1, n1 = acc.intersect(_rays(pkg, rays))
    _assert_hits_equal(h0, n0, ref, "linear leaves")
    _assert_hits_equal(h1, n1, ref, f"accelerated leaves (sub_leaf={sub_leaf})")
    assert ref["hit"].sum() > 0.3 * len(rays)


def test_leaf_accelerator_ties_and_on_plane(pkg, orc):
    """Coplanar, overlapping, duplicated triangles in ONE fat leaf: equal-t ties must go to the earliest scan
    position and origin-on-plane acceptances to the last one, whatever order the accelerator visits them in."""
    rng = np.random.RandomState(21)
    n = 40000  # 2048 leaves of ~20 triangles; few distinct planes and shapes so that ties are everywhere
    planes_z = np.float32([0.0, 0.25, 0.5])
    rows, tris = [], []
    base_tris = rng.uniform(-1, 1, (40, 3, 2)).astype(np.float32)
    for i in range(n):
        bt = base_tris[rng.randint(0, len(base_tris))]  # heavy duplication
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
    dd[m // 2:, 0:2] = rng.uniform(-0.3, 0.3, (m - m // 2, 2))
    rays = np.zeros((m, 7), np.float32)
    rays[:, 0:3], rays[:, 3:6], rays[:, 6] = oo, dd, FMAX
    rays[::9, 6] = rng.choice([0.0, 0.1, 0.25, 1.0, -3.0], len(rays[::9]))
    ref = o.intersect(rays)
    sc = pkg.Scene(sd)
    assert sc.num_subnodes() > 0
    hits, normals = sc.intersect(_rays(pkg, rays))
    _assert_hits_equal(hits, normals, ref, "ties/on-plane")
    assert (ref["t"][ref["hit"] == 1] == 0).sum() > 500  # origin-on-plane acceptances happened


# ---------------------------------------------------------------------------------------------------
# dragon-scale: BASELINE.json full sizes, through size-independent properties + sampled oracle check
# ---------------------------------------------------------------------------------------------------
def test_dragon_87k_full_frame_vs_oracle(pkg, orc):
    """87 K triangles (the report's dragon size), 640x360 frame, every ray against the oracle."""
    sd = pkg.scenes.make_dragon(87_000)
    W, H = 640, 360
    cam = pkg.scenes.default_camera(W, H)
    sc, o = pkg.Scene(sd), orc.OracleScene(sd)
    hits, normals = sc.trace_primary(cam, W, H, want_normals=True)
    _assert_hits_equal(hits, normals, o.intersect(sc.generate_rays(cam, W, H)), "dragon87k")


def test_config5_frame_on_one_gpu(pkg, orc):
    """BASELINE.json config 5's frame (3840x2160, 800 K triangles) on ONE GPU: all 8.3 M pixels against the oracle
    (flag, t bits, primitive id, material, normal bits), and its two-rank tiling merged into the same bytes."""
    sd = pkg.scenes.make_dragon(800_000)
    W, H = 3840, 2160
    cam = pkg.scenes.default_camera(W, H)
    sc = pkg.Scene(sd)
    hits, normals = sc.trace_primary(cam, W, H, want_normals=True)
    ref = orc.OracleScene(sd).intersect(sc.generate_rays(cam, W, H))
    _assert_hits_equal(hits, normals, ref, "dragon800k 4K")
    assert (hits["hit"] == 1).sum() > 500_000
    merged = np.zeros_like(hits)
    for r in range(2):
        part, _ = sc.trace_primary(cam, W, H, rank=r, nranks=2)
        own = pkg.tiling.owned_mask(W, H, r, 2).reshape(-1)
        merged[own] = part[own]
    assert merged.tobytes() == hits.tobytes()


def test_dragon_800k_1080p_properties(pkg, orc):
    """Config 4 at full size.  Checked: (1) every 16th row against the oracle; (2) determinism;
    (3) tile partition invariance; (4) every reported hit re-verified by the element-wise triangle primitive:
    the reported triangle is hit at exactly the reported t, and nothing reports t beyond FLT_MAX."""
    sd = pkg.scenes.make_dragon(800_000)
    W, H = 1920, 1080
    cam = pkg.scenes.default_camera(W, H)
    sc = pkg.Scene(sd)
    assert sc.num_levels() == 12
    hits, _ = sc.trace_primary(cam, W, H)
    again, _ = sc.trace_primary(cam, W, H)
    assert hits.tobytes() == again.tobytes()
    o = orc.OracleScene(sd)
    rows = np.arange(0, H, 16)
    idx = (rows[:, None] * W + np.arange(W)[None, :]).reshape(-1)
    rays = sc.generate_rays(cam, W, H)
    ref = o.intersect(rays[idx])
    _assert_hits_equal(hits[idx], None, ref, "dragon800k rows")
    m = hits["hit"] == 1
    assert 0.1 < m.mean() < 0.9
    # re-verify every hit with the primitive on the reported triangle
    prim = hits["prim_id"][m]
    v = sd.pos_nrm[sd.tri[prim].reshape(-1)].reshape(-1, 3, 6)
    tri18 = np.concatenate([v[:, :, 0:3].reshape(-1, 9), v[:, :, 3:6].reshape(-1, 9)], 1)
    t, hit, _ = pkg.ray_triangle(tri18, rays[m])
    assert hit.all() and np.array_equal(bits(t), bits(hits["t"][m]))
    part = [sc.trace_primary(cam, W, H, rank=r, nranks=2)[0] for r in range(2)]
    own0 = ~np.isnan(part[0]["t"])
    merged = np.where(own0, part[0], part[1])
    assert merged.tobytes() == hits.tobytes()


@pytest.mark.parametrize("view", ["default", "zoomed"])
def test_dodge_full_frame_800x800_primary_and_depth2(pkg, orc, scene_data, view):
    """The largest real mesh the reference ships (data/dodgeColorTest.obj: 16 311 triangles in 11 meshes, arrays committed as
    tests/golden/scenes/dodge.npz, un-normalised like the reference's Custom preset loads a model, scene.cpp:57-60) at the
    reference's own window size (windowResolution{800, 800}, main.cpp:29), every pixel, from the default camera (main.cpp:730-731:
    the car covers 3.5 % of the frame) and from the same trackball zoomed onto the car (distance 1, look-at = its centre: 25 %):
    primary hits bit for bit (flag, t bits, primitive, material, normal bits) against the oracle in both kernel shapes and with the
    exact walk, and the depth-2 frame (the reference's recursion cap, main.cpp:267) RGB within 1e-5 with equal ray counts."""
    sd = scene_data("dodge")
    W = H = 800
    cam = np.asarray(pkg.scenes.default_camera(W, H), np.float32).copy()
    if view == "zoomed":
        cam[0:3] = [0.0712, 0.0169, 0.2369]
        cam[6] = 1.0
    o = orc.OracleScene(sd)
    sc = pkg.Scene(sd)
    rays = sc.generate_rays(cam, W, H)
    assert np.array_equal(rays.view(np.float32).reshape(-1, 7).view(np.uint32), orc.generate_rays(cam, W, H).view(np.uint32))
    ref = o.intersect(rays)
    assert (0.02 if view == "default" else 0.15) < (ref["hit"] == 1).mean() < 0.95 and len(np.unique(ref["material"][ref["hit"] == 1])) >= 3
    for shape in (0, 1):
        pkg.set_kernel_shape(shape)
        try:
            hits, normals = sc.trace_primary(cam, W, H, want_normals=True)
        finally:
            pkg.set_kernel_shape(-1)
        _assert_hits_equal(hits, normals, ref, f"dodge 800x800 {view} primary, kernel shape {shape}")
    if sc.walk():
        sc.set_walk(False)
        hits, normals = sc.trace_primary(cam, W, H, want_normals=True)
        _assert_hits_equal(hits, normals, ref, f"dodge 800x800 {view} primary, exact walk")
        sc.set_walk(True)
    rgb, st = sc.render(cam, W, H, max_level=2)
    refrgb, nrays = o.render(cam, W, H, sd.point_lights, max_level=2)
    err = np.abs(rgb.astype(np.float64) - refrgb).max()
    assert err <= 1e-5, f"max abs RGB error {err}"  # BASELINE.json north_star: final pixel RGB within 1e-5 abs
    assert st["primary_rays"] + st["shadow_rays"] + st["reflection_rays"] == nrays and st["primary_rays"] == W * H
    assert (refrgb.sum(1) > 0).mean() > 0.02
