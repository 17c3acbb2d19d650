"""GPU tests of cgrt_render's predicted frames (capi.cpp render_impl; include/cgrt.h cgrt_set_render_prediction).
A frame of the same shape as the scene's previous frame is issued in one go -- every launch sized from what the previous frame found
per level, the kernels stopping at the counts on the device -- and checked once behind the frame; a frame whose lists outgrew their
launches, or that reaches a level the previous frame did not, is drawn again the exact way.  Same kernels on the same lists, so the
pixels and the ray counts must equal the exact frame's bit for bit (and through it agree with the oracle as the exact frame does,
tests/test_host_mirror.py), whatever the camera does between frames."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EXACT, PREDICTED, REDRAWN = 0, 1, 2
KEYS = ("primary_rays", "shadow_rays", "reflection_rays", "levels")


def _camera(pkg, W, H, distance=3.0, yaw_deg=20.0):
    cam = pkg.scenes.default_camera(W, H).copy()
    cam[4] = np.float32(yaw_deg) * np.float32(0.01745329251994329576923690768489)
    cam[6] = distance
    return cam


@pytest.fixture()
def exact_frames(pkg):
    """render(scene, ...) with prediction switched off around the call."""

    def render(sc, *a, **k):
        pkg.set_render_prediction(False)
        try:
            return sc.render(*a, **k)
        finally:
            pkg.set_render_prediction(True)

    return render


@pytest.mark.parametrize("scene", ["cornell", "monkey", "dragon"])
@pytest.mark.parametrize("depth", [1, 2, 4])
def test_predicted_frames_equal_exact_frames(pkg, orc, scene_data, exact_frames, scene, depth):
    sd = pkg.scenes.make_dragon(40_000) if scene == "dragon" else scene_data(scene)
    W, H = 320, 200
    cam = _camera(pkg, W, H)
    sc = pkg.Scene(sd)
    ref, st_ref = exact_frames(sc, cam, W, H, max_level=depth)
    assert sc.last_render_path() == EXACT
    first, st0 = sc.render(cam, W, H, max_level=depth)  # nothing to predict from (the frame above did not record any)
    assert sc.last_render_path() == EXACT and first.tobytes() == ref.tobytes()
    for _ in range(3):
        rgb, st = sc.render(cam, W, H, max_level=depth)
        assert sc.last_render_path() == PREDICTED
        assert rgb.tobytes() == ref.tobytes()
        for k in KEYS:
            assert st[k] == st_ref[k], k
    assert np.count_nonzero(ref) > 0
    if scene != "dragon":  # and the oracle's recursive per-pixel driver, on the predicted frame itself (shading: within 1e-5, powf;
        # the rule of tests/test_host_mirror.py::test_device_render_matches_oracle), ray counts exactly
        o_rgb, o_rays = orc.OracleScene(sd).render(cam, W, H, sd.point_lights, max_level=depth)
        assert np.abs(rgb.astype(np.float64) - o_rgb).max() <= 1e-5
        assert o_rays == st["primary_rays"] + st["shadow_rays"] + st["reflection_rays"]
    # another shape of frame starts from scratch: size, depth
    sc.render(cam, W // 2, H, max_level=depth)
    assert sc.last_render_path() == EXACT
    sc.render(cam, W, H, max_level=depth + 1)
    assert sc.last_render_path() == EXACT
    sc.render(cam, W, H, max_level=depth + 1)
    assert sc.last_render_path() == PREDICTED


def test_a_moving_camera_is_followed_and_a_jump_is_redrawn(pkg, scene_data, exact_frames):
    sd = scene_data("cornell")
    W, H = 400, 300
    sc, sc_ref = pkg.Scene(sd), pkg.Scene(sd)
    paths = []
    # a slow orbit: every frame is within the margin of the one before
    for k in range(8):
        cam = _camera(pkg, W, H, distance=3.0, yaw_deg=20.0 + 1.5 * k)
        rgb, st = sc.render(cam, W, H, max_level=4)
        ref, st_ref = exact_frames(sc_ref, cam, W, H, max_level=4)
        paths.append(sc.last_render_path())
        assert rgb.tobytes() == ref.tobytes(), k
        assert all(st[x] == st_ref[x] for x in KEYS), k
    assert paths[0] == EXACT and paths.count(PREDICTED) >= 5, paths
    # far away (few hits), then close up (the hit count grows several times over): the close frame cannot fit the far frame's launches
    far, near = _camera(pkg, W, H, distance=12.0), _camera(pkg, W, H, distance=2.0)
    sc.render(far, W, H, max_level=4)
    sc.render(far, W, H, max_level=4)
    assert sc.last_render_path() == PREDICTED
    rgb, st = sc.render(near, W, H, max_level=4)
    assert sc.last_render_path() == REDRAWN
    ref, st_ref = exact_frames(sc_ref, near, W, H, max_level=4)
    assert rgb.tobytes() == ref.tobytes() and all(st[x] == st_ref[x] for x in KEYS)
    assert st["primary_rays"] == W * H and st["shadow_rays"] > 4 * sc.render(far, W, H, max_level=4)[1]["shadow_rays"]
    # ... and the other way round (the lists shrink: nothing to redraw)
    assert sc.last_render_path() == REDRAWN or sc.last_render_path() == PREDICTED
    rgb, _ = sc.render(far, W, H, max_level=4)
    assert sc.last_render_path() == PREDICTED
    assert rgb.tobytes() == exact_frames(sc_ref, far, W, H, max_level=4)[0].tobytes()


def test_a_frame_that_reaches_a_deeper_level_is_redrawn(pkg, scene_data, exact_frames):
    """Depth 4 on the mirror Cornell box: looking away from the mirror the frame ends after level 0 or 1; turning towards it brings
    levels the previous frame did not issue."""
    sd = scene_data("cornell")
    W, H = 320, 240
    sc, sc_ref = pkg.Scene(sd), pkg.Scene(sd)
    levels = {}
    for yaw in range(0, 360, 30):
        cam = _camera(pkg, W, H, distance=3.0, yaw_deg=float(yaw))
        _, st = exact_frames(sc_ref, cam, W, H, max_level=4)
        levels[yaw] = st["levels"]
    lo, hi = min(levels, key=levels.get), max(levels, key=levels.get)
    if levels[lo] == levels[hi]:
        pytest.skip(f"every view of this scene reaches {levels[hi]} levels")
    a, b = _camera(pkg, W, H, yaw_deg=float(lo)), _camera(pkg, W, H, yaw_deg=float(hi))
    sc.render(a, W, H, max_level=4)
    sc.render(a, W, H, max_level=4)
    assert sc.last_render_path() == PREDICTED
    rgb, st = sc.render(b, W, H, max_level=4)
    assert sc.last_render_path() in (REDRAWN, PREDICTED)  # (PREDICTED only if the deeper view also fits -- it cannot: a level is missing)
    assert sc.last_render_path() == REDRAWN
    ref, st_ref = exact_frames(sc_ref, b, W, H, max_level=4)
    assert rgb.tobytes() == ref.tobytes() and st["levels"] == st_ref["levels"] == levels[hi]


def test_rank_tiles_predict_per_rank(pkg, scene_data):
    sd = scene_data("monkey")
    W, H = 256, 192
    cam = _camera(pkg, W, H)
    whole, _ = pkg.Scene(sd).render(cam, W, H, max_level=2)
    scs = [pkg.Scene(sd) for _ in range(3)]
    for rep in range(3):
        frame = np.zeros((W * H, 3), np.float32)
        for r, sc in enumerate(scs):
            sc.render_rank(cam, W, H, r, 3, rgb=frame, max_level=2)
            assert sc.last_render_path() == (EXACT if rep == 0 else PREDICTED)
        assert frame.tobytes() == whole.tobytes()
