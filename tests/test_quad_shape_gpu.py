"""GPU parity tests of the kernel SHAPES (trace_kernels.hip "kernel shape per launch"): how a launch lays its rays out on the lanes.
  0 lane64  one ray per lane, 64 per wave (round 2's shape)         2 lane16  one ray per lane, 16 per single-wave workgroup
  1 quad16  quad per ray, a node's four boxes in parallel (walk_quad.h), 16 rays per wave, 16 lanes per ray once <= 4 are live
  3 quad4   4 rays per wave: 16 lanes per ray (4 stack entries x 4 boxes) from the first step
The shape changes only the layout of the certified search -- same tree, same arithmetic, same scan rule, same certificate, exact
walk as fallback -- so every result must equal lane64's, hence the oracle's, bit for bit: ray lists (closest hit + normals),
occlusion lists inside cgrt_render (incl. the choice made on the device for lists sized there), primary frames (plain,
rank-tiled, packed multi-device order) and whole shaded frames."""
import numpy as np
import pytest

import rayfam
from conftest import bits
from test_parity_gpu import _assert_hits_equal, _rays

pytestmark = pytest.mark.gpu


@pytest.fixture()
def shapes(pkg):
    """Calls fn under shapes 0 (lane64) and `others` (default: quad16) and returns the results in that order."""

    def both(fn, others=(1,)):
        out = []
        for mode in (0,) + tuple(others):
            pkg.set_kernel_shape(mode)
            try:
                out.append(fn())
            finally:
                pkg.set_kernel_shape(-1)
        return out

    return both


@pytest.fixture()
def forced_fast_tree(pkg):
    pkg.set_fast_tree(1)
    yield
    pkg.set_fast_tree(-1)


@pytest.mark.parametrize("ntris", [20_000, 87_000])
def test_quad_shape_ray_lists_equal_lane_shape_and_oracle(pkg, orc, shapes, ntris):
    sd = pkg.scenes.make_dragon(ntris)
    o = orc.OracleScene(sd)
    _, boxes = o.nodes()
    W = H = 128
    cam = pkg.scenes.default_camera(W, H)
    fam = rayfam.families(sd, boxes, orc.generate_rays(cam, W, H), rng=np.random.RandomState(31), n_random=4000)
    sc = pkg.Scene(sd)
    assert sc.walk() == 1
    for k in sorted(fam):
        r = _rays(pkg, fam[k])
        res = shapes(lambda: sc.intersect(r), others=(1, 2, 3))
        for h, nn in res[1:]:
            assert res[0][0].tobytes() == h.tobytes() and res[0][1].tobytes() == nn.tobytes(), k
        _assert_hits_equal(res[1][0], res[1][1], o.intersect(fam[k]), f"dragon{ntris}/{k} quad shape")
    # list lengths that are not multiples of 16 / 4 (a wave's rays): partial waves, single rays
    rays = _rays(pkg, rayfam.concat(fam))
    for n in (1, 3, 5, 15, 17, 63, 65, 1000):
        res = shapes(lambda: sc.intersect(rays[:n]), others=(1, 2, 3))
        for h, nn in res[1:]:
            assert res[0][0].tobytes() == h.tobytes() and res[0][1].tobytes() == nn.tobytes(), n
    # the default policy (by size) on lists either side of its thresholds
    pkg.set_kernel_shape(-1)
    big = np.tile(rays, 1 + 140_000 // len(rays))
    ref_h, ref_n = res[0][0], res[0][1]
    for n in (8192, 8193, 131_072, 131_073):
        h, nn = sc.intersect(big[:n])
        pkg.set_kernel_shape(0)
        h0, n0 = sc.intersect(big[:n])
        pkg.set_kernel_shape(-1)
        assert h.tobytes() == h0.tobytes() and nn.tobytes() == n0.tobytes(), n
    # the work the two shapes report: same rays enter the tree, same rays fall back; the wide tail steps speculatively, so
    # node visits may differ a little, never the answer
    c0, c1 = shapes(lambda: sc.count_batch(rays))
    assert c0["rays"] == c1["rays"] == len(rays) and c0["tree_rays"] == c1["tree_rays"] > 0
    assert c0["fallback_rays"] == c1["fallback_rays"] and c0["cert_boxes"] == c1["cert_boxes"]
    assert 0.8 * c0["sub_visits"] <= c1["sub_visits"] <= 1.3 * c0["sub_visits"], (c0, c1)


@pytest.mark.parametrize("name", ["cube", "cornell", "monkey", "dodge", "blob"])
def test_quad_shape_on_thin_leaf_scenes(pkg, orc, scene_data, shapes, forced_fast_tree, name):
    """Zero-thickness boxes, ties on shared edges, certificates that fail all the time: the fallback path of the quad shape
    (one lane per quad walks exactly) gets real traffic here."""
    sd = scene_data(name)
    o = orc.OracleScene(sd)
    _, boxes = o.nodes()
    W = H = 96
    cam = pkg.scenes.default_camera(W, H)
    fam = rayfam.families(sd, boxes, orc.generate_rays(cam, W, H), rng=np.random.RandomState(5), n_random=3000)
    sc = pkg.Scene(sd)
    assert sc.walk() == 1
    rays = rayfam.concat(fam)
    res = shapes(lambda: sc.intersect(_rays(pkg, rays)), others=(1, 2, 3))
    (h0, n0), (h1, n1) = res[0], res[1]
    for h, nn in res[1:]:
        assert h0.tobytes() == h.tobytes() and n0.tobytes() == nn.tobytes()
    _assert_hits_equal(h1, n1, o.intersect(rays), f"{name} quad shape")
    c0, c1 = shapes(lambda: sc.count_batch(_rays(pkg, rays)))
    assert c0["fallback_rays"] == c1["fallback_rays"]
    if name in ("cube", "cornell"):
        assert c1["fallback_rays"] > 0


def test_quad_shape_primary_frames(pkg, orc, shapes):
    sd = pkg.scenes.make_dragon(60_000)
    W, H = 500, 301  # not a multiple of the 8x8 tiles nor of the 64x64 super-tiles
    cam = pkg.scenes.default_camera(W, H)
    sc = pkg.Scene(sd)
    (h0, n0), (h1, n1) = shapes(lambda: sc.trace_primary(cam, W, H, want_normals=True))
    assert h0.tobytes() == h1.tobytes() and n0.tobytes() == n1.tobytes()
    ref = orc.OracleScene(sd).intersect(sc.generate_rays(cam, W, H))
    _assert_hits_equal(h1, n1, ref, "quad-shape primary frame")
    c0, c1 = shapes(lambda: sc.count_primary(cam, W, H))
    assert c0["rays"] == c1["rays"] == W * H and c0["tree_rays"] == c1["tree_rays"] and c0["fallback_rays"] == c1["fallback_rays"]
    # rank tiles (image tiling across GPUs) and a sub-rectangle
    for r in range(3):
        (p0, _), (p1, _) = shapes(lambda: sc.trace_primary(cam, W, H, rank=r, nranks=3))
        assert p0.tobytes() == p1.tobytes()
    (q0, _), (q1, _) = shapes(lambda: sc.trace_primary(cam, W, H, rect=(37, 11, 403, 290)))
    assert q0.tobytes() == q1.tobytes()
    # packed order of the multi-device entry (FrameDev::packed): the quad launch must write where the lane launch writes
    sc2 = pkg.Scene(sd)
    (m0, mn0, _), (m1, mn1, _) = shapes(lambda: pkg.trace_primary_multi([sc, sc2], cam, W, H, want_normals=True))
    assert m0.tobytes() == m1.tobytes() == h0.tobytes()
    hit = h0["hit"] == 1
    assert mn0[hit].tobytes() == mn1[hit].tobytes() == n0[hit].tobytes()


@pytest.mark.parametrize("scene", ["dragon", "cornell"])
def test_quad_shape_shaded_frames(pkg, scene_data, shapes, forced_fast_tree, scene):
    """cgrt_render: compact primary kernel, occlusion lists (k_trace_shadow) and mirror lists in the quad shape."""
    sd = pkg.scenes.make_dragon(40_000) if scene == "dragon" else scene_data("cornell")
    W, H = 320, 200
    cam = pkg.scenes.default_camera(W, H)
    sc = pkg.Scene(sd)
    for depth in (1, 2, 4):
        res = shapes(lambda: sc.render(cam, W, H, max_level=depth), others=(1, 2, 3, -1))  # -1: the lists pick 16 / 64 per wave on the device
        rgb0, st0 = res[0]
        for rgb1, st1 in res[1:]:
            assert rgb0.tobytes() == rgb1.tobytes(), depth
            for k in ("primary_rays", "shadow_rays", "reflection_rays", "levels"):
                assert st0[k] == st1[k], (k, depth)
    assert np.count_nonzero(rgb0) > 0


def test_kernel_shape_api(pkg):
    mode, max_rays = pkg.kernel_shape()
    assert mode == -1 and max_rays > 0
    pkg.set_kernel_shape(1)
    assert pkg.kernel_shape()[0] == 1
    pkg.set_kernel_shape(-1, 12345)
    assert pkg.kernel_shape() == (-1, 12345)
    pkg.set_kernel_shape(-1, max_rays)
    for m in (0, 1, 2, 3, -1):
        pkg.set_kernel_shape(m)
        assert pkg.kernel_shape()[0] == m
    with pytest.raises(pkg.CgrtError):
        pkg.set_kernel_shape(7)
