"""GPU parity tests of the QUAD-PER-RAY kernel shape (walk_quad.h): four lanes per ray, a node's four boxes in parallel, 16
rays per wave, a row of 16 lanes per ray once <= 4 rays are live.  The shape changes only how the certified search is laid
out on the lanes -- same tree, same arithmetic, same scan rule, same certificate, exact walk as fallback -- so every result must
equal the lane-per-ray shape's, hence the oracle's, bit for bit: ray lists (closest hit + normals), occlusion lists inside
cgrt_render, primary frames (plain, rank-tiled, packed multi-device order) and whole shaded frames."""
import numpy as np
import pytest

import rayfam
from conftest import bits
from test_parity_gpu import _assert_hits_equal, _rays

pytestmark = pytest.mark.gpu


@pytest.fixture()
def shapes(pkg):
    """Calls fn under both shapes and returns (lane_per_ray, quad_per_ray)."""

    def both(fn):
        out = []
        for mode in (0, 1):
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
        (h0, n0), (h1, n1) = shapes(lambda: sc.intersect(r))
        assert h0.tobytes() == h1.tobytes() and n0.tobytes() == n1.tobytes(), k
        _assert_hits_equal(h1, n1, o.intersect(fam[k]), f"dragon{ntris}/{k} quad shape")
    # list lengths that are not multiples of 16 (a wave's rays) or 4: partial waves, single rays
    rays = _rays(pkg, rayfam.concat(fam))
    for n in (1, 3, 15, 17, 63, 65, 1000):
        (h0, n0), (h1, n1) = shapes(lambda: sc.intersect(rays[:n]))
        assert h0.tobytes() == h1.tobytes() and n0.tobytes() == n1.tobytes(), n
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
    (h0, n0), (h1, n1) = shapes(lambda: sc.intersect(_rays(pkg, rays)))
    assert h0.tobytes() == h1.tobytes() and n0.tobytes() == n1.tobytes()
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
        (rgb0, st0), (rgb1, st1) = shapes(lambda: sc.render(cam, W, H, max_level=depth))
        assert rgb0.tobytes() == rgb1.tobytes(), depth
        for k in ("primary_rays", "shadow_rays", "reflection_rays", "levels"):
            assert st0[k] == st1[k], (k, depth)
    assert np.count_nonzero(rgb1) > 0


def test_kernel_shape_api(pkg):
    mode, max_rays = pkg.kernel_shape()
    assert mode == -1 and max_rays > 0
    pkg.set_kernel_shape(1)
    assert pkg.kernel_shape()[0] == 1
    pkg.set_kernel_shape(-1, 12345)
    assert pkg.kernel_shape() == (-1, 12345)
    pkg.set_kernel_shape(-1, max_rays)
    with pytest.raises(pkg.CgrtError):
        pkg.set_kernel_shape(7)
