"""Soft shadows of spherical lights (src/main.cpp:168-218; SceneType::CornellBoxSphericalLight, scene.cpp:27-32).

Upstream draws every sample direction from std::random_device (main.cpp:46-59), so no two runs of the reference agree
with each other, let alone with anything else.  The library takes the draws as DATA (a table of unit vectors + an
integer hash that picks from it, include/cgrt.h) and parity is defined in two ways:
  * same table, same seed  -> device frame within 1e-5 of the oracle's per-pixel recursive restatement;
  * different tables       -> the frames agree statistically (what "parity" can mean against upstream itself)."""
import ctypes as C

import numpy as np
import pytest


def _soft(pkg):
    return pkg.scenes.CORNELL_SPHERICAL_LIGHTS.copy()


def test_unit_vector_table_is_unit_and_isotropic(pkg):
    u = pkg.unit_vector_table(1 << 14, seed=3)
    assert u.dtype == np.float32 and u.shape == (1 << 14, 3)
    assert np.abs(np.linalg.norm(u.astype(np.float64), axis=1) - 1).max() < 1e-6
    assert np.abs(u.mean(0)).max() < 0.03  # isotropic: mean ~ 0 (sigma = 1/sqrt(3 n) = 0.0045)
    assert np.array_equal(u, pkg.unit_vector_table(1 << 14, seed=3))


def test_oracle_soft_radius_zero_is_a_point_light(pkg, orc, scene_data):
    """A spherical light of radius 0 sends every sample to its centre: the frame equals the hard-shadow frame of a
    point light there (up to the two drivers' different epsilon rules, which only matter for blockers within 0.001
    of the light: none in the Cornell box)."""
    sd = scene_data("cornell")
    W, H = 48, 32
    cam = pkg.scenes.default_camera(W, H)
    o = orc.OracleScene(sd)
    sl = np.asarray([[0, 0.58, 0, 0.0, 1, 1, 1]], np.float32)
    soft, n_soft = o.render_soft(cam, W, H, np.zeros((0, 6), np.float32), sl, pkg.unit_vector_table(64, 1), samples=3, max_level=2)
    hard, n_hard = o.render(cam, W, H, np.asarray([[0, 0.58, 0, 1, 1, 1]], np.float32), max_level=2)
    assert np.abs(soft - hard).max() <= 1e-6
    assert n_soft > n_hard  # 3 samples per hit instead of 1 shadow ray


def test_oracle_soft_is_deterministic_and_statistically_stable(pkg, orc, scene_data):
    sd = scene_data("cornell")
    W, H = 48, 32
    cam = pkg.scenes.default_camera(W, H)
    o = orc.OracleScene(sd)
    none = np.zeros((0, 6), np.float32)
    a, na = o.render_soft(cam, W, H, none, _soft(pkg), pkg.unit_vector_table(4096, 1), samples=64, seed=5)
    a2, _ = o.render_soft(cam, W, H, none, _soft(pkg), pkg.unit_vector_table(4096, 1), samples=64, seed=5, threads=1)
    b, nb = o.render_soft(cam, W, H, none, _soft(pkg), pkg.unit_vector_table(4096, 2), samples=64, seed=9)
    assert np.array_equal(a, a2) and na == nb
    assert not np.array_equal(a, b)  # penumbra pixels differ draw by draw...
    assert np.abs(a - b).mean() < 0.01 and np.abs(a.mean() - b.mean()) < 0.003  # ...but the frames agree statistically
    # the soft frame has partially lit pixels, the thing hard shadows cannot produce
    full, _ = o.render_soft(cam, W, H, none, np.asarray([[0, 0.45, 0, 0.0, 1, 1, 1]], np.float32), pkg.unit_vector_table(64, 1), samples=1)
    lit = (full.sum(1) > 1e-3)
    frac = a.sum(1)[lit] / full.sum(1)[lit]
    assert ((frac > 0.05) & (frac < 0.95)).sum() > 0


def test_render_soft_rejects_bad_arguments(pkg, scene_data):
    s = pkg.Scene(scene_data("cube"), device=-1)
    cam = pkg.Camera.from_array(pkg.scenes.default_camera(8, 8))
    rgb = np.zeros(8 * 8 * 3, np.float32)
    sl = _soft(pkg)
    u = pkg.unit_vector_table(16, 0)
    q = pkg.SoftShadows(sl.ctypes.data, u.ctypes.data, 1, 4, 16, 0, 0)
    rc = pkg.lib().cgrt_render_soft(s._h, C.byref(cam), 8, 8, None, 0, C.byref(q), 2, rgb.ctypes.data_as(C.c_void_p), None)
    assert rc == -2  # host-only scene: there is no CPU path
    assert b"no CPU traversal path" in pkg.lib().cgrt_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("name,W,H,level,samples,with_point", [("cornell", 96, 64, 2, 16, False), ("cornell", 96, 64, 3, 5, True),
                                                                ("cornell", 64, 48, 2, 200, False), ("blob", 80, 60, 2, 70, True),
                                                                ("cornell", 64, 64, 1, 64, False)])
def test_device_soft_shadows_match_oracle(pkg, orc, scene_data, name, W, H, level, samples, with_point):
    """cgrt_render_soft == the oracle's restatement of main.cpp:168-218 fed the same draws: RGB within 1e-5; the
    any-hit early exit of the sample rays changes nothing (byte-identical to the closest-hit walk); ray counts match."""
    sd = scene_data(name)
    cam = pkg.scenes.default_camera(W, H)
    sl = _soft(pkg) if name == "cornell" else np.asarray([[-0.6, 0.9, -0.7, 0.15, 1, 0.9, 0.8], [0.8, 0.7, -0.9, 0.05, 0.3, 0.3, 0.5]], np.float32)
    pl = sd.point_lights if with_point else np.zeros((0, 6), np.float32)
    units = pkg.unit_vector_table(4096, seed=11)
    s = pkg.Scene(sd)
    rgb, st = s.render_soft(cam, W, H, sl, units, samples=samples, seed=7, lights=pl, max_level=level)
    rgb_c, st_c = s.render_soft(cam, W, H, sl, units, samples=samples, seed=7, lights=pl, max_level=level, closest_hit=True)
    ref, nrays = orc.OracleScene(sd).render_soft(cam, W, H, pl, sl, units, samples=samples, seed=7, max_level=level)
    err = np.abs(rgb.astype(np.float64) - ref).max()
    assert err <= 1e-5, f"max abs RGB error {err}"
    assert np.array_equal(rgb.view(np.uint32), rgb_c.view(np.uint32))
    assert st["primary_rays"] + st["shadow_rays"] + st["reflection_rays"] + st["soft_shadow_rays"] == nrays
    assert st["soft_shadow_rays"] > 0 and st["soft_shadow_rays"] % (samples * len(sl)) == 0
    assert (ref.sum(1) > 0).mean() > 0.05


@pytest.mark.gpu
def test_device_soft_shadows_statistical_parity(pkg, orc, scene_data):
    """Against upstream itself only statistics can agree: device frame with one set of draws vs oracle frame with
    another, 200 samples (main.cpp:176)."""
    sd = scene_data("cornell")
    W, H = 96, 64
    cam = pkg.scenes.default_camera(W, H)
    none = np.zeros((0, 6), np.float32)
    rgb, _ = pkg.Scene(sd).render_soft(cam, W, H, _soft(pkg), pkg.unit_vector_table(1 << 16, 21), samples=200, seed=1, lights=none)
    ref, _ = orc.OracleScene(sd).render_soft(cam, W, H, none, _soft(pkg), pkg.unit_vector_table(1 << 16, 22), samples=200, seed=2)
    d = np.abs(rgb.astype(np.float64) - ref)
    assert d.mean() < 0.004 and d.max() < 0.15  # a penumbra pixel: sigma = sqrt(p(1-p)/200) <= 0.035 per frame
    assert abs(rgb.mean() - ref.mean()) < 0.001


@pytest.mark.gpu
def test_device_soft_shadows_seed_and_table_matter(pkg, scene_data):
    sd = scene_data("cornell")
    W, H = 64, 48
    cam = pkg.scenes.default_camera(W, H)
    none = np.zeros((0, 6), np.float32)
    s = pkg.Scene(sd)
    u = pkg.unit_vector_table(2048, 5)
    a, _ = s.render_soft(cam, W, H, _soft(pkg), u, samples=32, seed=1, lights=none)
    a2, _ = s.render_soft(cam, W, H, _soft(pkg), u, samples=32, seed=1, lights=none)
    b, _ = s.render_soft(cam, W, H, _soft(pkg), u, samples=32, seed=2, lights=none)
    assert np.array_equal(a, a2) and not np.array_equal(a, b)
    # no spherical light: cgrt_render_soft == cgrt_render
    c, _ = s.render_soft(cam, W, H, np.zeros((0, 7), np.float32), u, lights=sd.point_lights)
    d, _ = s.render(cam, W, H)
    assert np.array_equal(c, d)


@pytest.mark.gpu
def test_host_mirror_soft_shadows_match_oracle(pkg, orc, scene_data):
    """The C++ mirror's renderRayTracing with Scene::sphericalLight (the reference's own scene field): the host-driven
    wavefront sends the sample rays through BoundingVolumeHierarchy::intersectBatch; same draws -> RGB within 1e-5,
    and == the device driver's frame to the same bar."""
    sd = scene_data("cornell")
    W, H = 80, 60
    cam = pkg.scenes.default_camera(W, H)
    units = pkg.unit_vector_table(4096, seed=4)
    sl = _soft(pkg)
    rgb, st = pkg.host_render_soft(sd, cam, W, H, sl, units, samples=24, seed=3, max_level=3)
    ref, nrays = orc.OracleScene(sd).render_soft(cam, W, H, sd.point_lights, sl, units, samples=24, seed=3, max_level=3)
    dev, _ = pkg.Scene(sd).render_soft(cam, W, H, sl, units, samples=24, seed=3, max_level=3)
    assert np.abs(rgb.astype(np.float64) - ref).max() <= 1e-5
    assert np.abs(rgb.astype(np.float64) - dev).max() <= 1e-5
    assert st["primary"] + st["shadow"] + st["reflection"] + st["soft_shadow"] == nrays
    # the mirror's device-driven variant (renderToBufferOnDevice): the C++ surface over cgrt_render_soft
    rgb_d, st_d = pkg.host_render_soft(sd, cam, W, H, sl, units, samples=24, seed=3, max_level=3, on_device=True)
    assert np.array_equal(rgb_d, dev)
    assert st_d["primary"] + st_d["shadow"] + st_d["reflection"] + st_d["soft_shadow"] == nrays


@pytest.mark.gpu
def test_host_mirror_default_sampler_is_statistically_right(pkg, orc, scene_data):
    """SceneType::CornellBoxSphericalLight the reference's way: no table supplied, the mirror draws its own gaussian
    unit vectors (std::normal_distribution, like randomUnitVector()) -- agrees with the oracle statistically."""
    sd = scene_data("cornell")
    W, H = 64, 48
    cam = pkg.scenes.default_camera(W, H)
    none = np.zeros((0, 6), np.float32)
    rgb, st = pkg.host_render_soft(sd, cam, W, H, _soft(pkg), None, samples=200, seed=0, lights=none)
    ref, _ = orc.OracleScene(sd).render_soft(cam, W, H, none, _soft(pkg), pkg.unit_vector_table(1 << 16, 8), samples=200, seed=4)
    d = np.abs(rgb.astype(np.float64) - ref)
    assert d.mean() < 0.004 and abs(rgb.mean() - ref.mean()) < 0.001
    assert st["soft_shadow"] % 200 == 0 and st["soft_shadow"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 3])
def test_render_rank_frames_merge_into_the_single_rank_frame(pkg, scene_data, nranks):
    """Whole-frame image tiling across GPUs: each rank shades only its 64x64 super-tiles, keeps every
    secondary ray of its pixels, and leaves the other pixels alone; merged by ownership the ranks' frames are
    byte-identical to the single-rank frame (with point lights, a mirror bounce and soft shadows in play)."""
    sd = scene_data("cornell")
    W, H = 200, 136  # 4 x 3 super-tiles, ragged on both edges
    cam = pkg.scenes.default_camera(W, H)
    units = pkg.unit_vector_table(1024, 2)
    sl = _soft(pkg)
    s = pkg.Scene(sd)
    full, st_full = s.render_soft(cam, W, H, sl, units, samples=8, seed=3, max_level=3)
    merged = np.full((W * H, 3), -7.0, np.float32)
    owned = pkg.tiling.owned_mask(W, H, 0, nranks).reshape(-1)
    tot = dict(primary_rays=0, shadow_rays=0, reflection_rays=0, soft_shadow_rays=0)
    for r in range(nranks):
        before = merged.copy()
        _, st = s.render_rank(cam, W, H, r, nranks, rgb=merged, max_level=3, spherical=sl, units=units, samples=8, seed=3)
        mine = pkg.tiling.owned_mask(W, H, r, nranks).reshape(-1)
        assert np.array_equal(merged[~mine], before[~mine])  # other ranks' pixels untouched
        assert st["primary_rays"] == int(mine.sum())
        for k in tot:
            tot[k] += st[k]
    assert owned.any() and np.array_equal(merged.view(np.uint32), full.view(np.uint32))
    assert all(tot[k] == st_full[k] for k in tot)
