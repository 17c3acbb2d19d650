"""GPU tests of the primary frames' HINTS (cgrt_layout.h HintDev, capi.cpp attach_hints, include/cgrt.h cgrt_set_frame_hints): the
tiles whose wave took long in one frame are traced first (mode 1) or as four 16-ray waves (mode 2) by the next frame of the same
shape.  Only order and layout change, so every frame must equal the unhinted frame -- and through it the oracle -- bit for bit:
hits, t bits, primitive and material ids, normals; whole frames, rectangles, rank tiles, frames not divisible by the tile size,
a camera that moves, shapes that change between frames, two streams taking turns, several threads on one scene."""
import threading

import numpy as np
import pytest

from test_parity_gpu import _assert_hits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["default thresholds", "1 us", "20 ticks"])
def hints(pkg, request):
    """Sets the hint mode; parametrised over the thresholds: the shipped ones (few or no hard tiles in frames this small), 1 us
    (most tiles that reach the tree are hard) and 20 ticks (every tile that is traced at all: the lists overflow)."""
    thr = {"default thresholds": (0, 0), "1 us": (100, 60), "20 ticks": (20, 20)}[request.param]
    pkg.debug_set_hint_thresholds(*thr)

    def set_mode(mode):
        pkg.set_frame_hints(mode)

    yield set_mode
    pkg.set_frame_hints(-1)
    pkg.debug_set_hint_thresholds(0, 0)


def _camera(pkg, W, H, yaw_deg=20.0, distance=3.0):
    cam = pkg.scenes.default_camera(W, H).copy()
    cam[4] = np.float32(yaw_deg) * np.float32(0.01745329251994329576923690768489)
    cam[6] = distance
    return cam


@pytest.mark.parametrize("mode", [1, 2, -1])
@pytest.mark.parametrize("W,H", [(640, 360), (500, 301)])
def test_hinted_frames_equal_plain_frames_and_the_oracle(pkg, orc, hints, mode, W, H):
    sd = pkg.scenes.make_dragon(60_000)
    cam = _camera(pkg, W, H)
    sc = pkg.Scene(sd)
    hints(0)
    h0, n0 = sc.trace_primary(cam, W, H, want_normals=True)
    ref = orc.OracleScene(sd).intersect(sc.generate_rays(cam, W, H))
    _assert_hits_equal(h0, n0, ref, "plain frame")
    hints(mode)
    for k in range(12):  # two plain frames (a shape gets its buffers at the third), one that writes the first list, then frames that
        # read one list and write the next (three sets rotate)
        h, n = sc.trace_primary(cam, W, H, want_normals=True)
        assert h.tobytes() == h0.tobytes() and n.tobytes() == n0.tobytes(), (mode, k)
    # a rectangle and rank tiles: other shapes, new buffers, same pixels
    rect = (37, 11, W - 60, H - 9)
    hints(0)
    r0, _ = sc.trace_primary(cam, W, H, rect=rect)
    t0 = [sc.trace_primary(cam, W, H, rank=r, nranks=3)[0] for r in range(3)]
    hints(mode)
    for k in range(9):
        assert sc.trace_primary(cam, W, H, rect=rect)[0].tobytes() == r0.tobytes(), k
    for k in range(3):
        for r in range(3):  # the shape changes with every call: hints restart each time, never a wrong pixel
            assert sc.trace_primary(cam, W, H, rank=r, nranks=3)[0].tobytes() == t0[r].tobytes(), (k, r)
    for r in range(3):
        for k in range(9):  # and the same rank nine times in a row: its hints are used
            assert sc.trace_primary(cam, W, H, rank=r, nranks=3)[0].tobytes() == t0[r].tobytes(), (r, k)


@pytest.mark.parametrize("mode", [1, 2])
def test_hints_follow_a_moving_camera(pkg, hints, mode):
    """The hard list is one frame old: tiles that are hard no longer, tiles that are hard now and not listed, a model that moves out
    of the picture and back."""
    sd = pkg.scenes.make_dragon(120_000)
    W, H = 800, 450
    sc, sc_ref = pkg.Scene(sd), pkg.Scene(sd)
    cams = [_camera(pkg, W, H, yaw_deg=20.0 + 7.0 * k, distance=3.0 - 0.15 * (k % 5)) for k in range(10)]
    away = pkg.scenes.default_camera(W, H).copy()
    away[0:3] = [50.0, 0.0, 0.0]
    cams = cams[:5] + [away, away] + cams[5:]
    for k, cam in enumerate(cams):
        hints(0)
        ref = sc_ref.trace_primary(cam, W, H, want_normals=True)
        hints(mode)
        got = sc.trace_primary(cam, W, H, want_normals=True)
        assert got[0].tobytes() == ref[0].tobytes() and got[1].tobytes() == ref[1].tobytes(), k


def test_hints_on_thin_leaf_scenes_and_every_walk(pkg, scene_data, hints):
    """Scenes whose rays fall back to the exact walk all the time, a scene without a fast tree (no hints are attached), spheres."""
    W, H = 320, 240
    for name in ("cornell", "monkey", "cube", "spheres"):
        sd = scene_data(name)
        cam = pkg.scenes.default_camera(W, H)
        sc = pkg.Scene(sd)
        hints(0)
        h0, n0 = sc.trace_primary(cam, W, H, want_normals=True)
        for mode in (1, 2):
            hints(mode)
            for k in range(9):
                h, n = sc.trace_primary(cam, W, H, want_normals=True)
                assert h.tobytes() == h0.tobytes() and n.tobytes() == n0.tobytes(), (name, mode, k)


def test_hints_go_dormant_and_come_back(pkg, scene_data, hints):
    """A scene without long waves: after eight frames with empty lists the library traces 56 frames without hints, then probes
    again (capi.cpp attach_hints) -- 150 frames cross that cycle twice; and a threshold that adapts (HintDev::ctl) while the camera
    keeps moving."""
    sd = scene_data("monkey")
    W, H = 256, 192
    sc, sc_ref = pkg.Scene(sd), pkg.Scene(sd)
    hints(0)
    cams = [_camera(pkg, W, H, yaw_deg=20.0 + 3.0 * (k % 7)) for k in range(7)]
    refs = [sc_ref.trace_primary(c, W, H)[0].tobytes() for c in cams]
    for mode in (2, 1, -1):
        hints(mode)
        for k in range(150):
            assert sc.trace_primary(cams[k % 7], W, H)[0].tobytes() == refs[k % 7], (mode, k)


def test_two_streams_taking_turns_and_threads_on_one_scene(pkg, hints):
    """The hint buffers belong to the scene and assume frames that follow each other on one stream; callers that alternate between
    streams, or several threads rendering the same scene at once, must still get the right pixels (the library then leaves the hints
    out: cgrt.h)."""
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")  # (the runtime libcgrt.so is linked against: plain streams and buffers, no torch in the tests)

    def ok(rc):
        assert rc == 0, rc

    sd = pkg.scenes.make_dragon(60_000)
    W, H = 640, 360
    cam = _camera(pkg, W, H)
    sc = pkg.Scene(sd)
    hints(0)
    h0, _ = sc.trace_primary(cam, W, H)
    hints(1)
    nbytes = W * H * 16
    streams, bufs = [], []
    for _ in range(2):
        st, buf = C.c_void_p(), C.c_void_p()
        ok(hip.hipStreamCreateWithFlags(C.byref(st), 1))  # hipStreamNonBlocking
        ok(hip.hipMalloc(C.byref(buf), C.c_size_t(nbytes)))
        streams.append(st)
        bufs.append(buf)

    def frame_of(i):
        out = np.empty(W * H, pkg.HIT_DTYPE)
        ok(hip.hipDeviceSynchronize())
        ok(hip.hipMemcpy(C.c_void_p(out.ctypes.data), bufs[i], C.c_size_t(nbytes), 2))  # device to host
        return out

    order = [0] * 8 + [1, 0, 1] + [1] * 14 + [0] * 14 + [1]
    for k, i in enumerate(order):
        ok(hip.hipMemset(bufs[i], 0, C.c_size_t(nbytes)))
        ok(hip.hipDeviceSynchronize())
        sc.trace_primary_device(cam, W, H, bufs[i].value, stream=streams[i].value)
        assert frame_of(i).tobytes() == h0.tobytes(), (k, i)
    # back-to-back launches without a host sync in between, then one check
    ok(hip.hipMemset(bufs[0], 0, C.c_size_t(nbytes)))
    ok(hip.hipDeviceSynchronize())
    for k in range(12):
        sc.trace_primary_device(cam, W, H, bufs[0].value, stream=streams[0].value)
    assert frame_of(0).tobytes() == h0.tobytes()
    # ... and alternating streams without a host sync: the library must order (or leave out) what shares the hint buffers
    for k in range(16):
        sc.trace_primary_device(cam, W, H, bufs[k & 1].value, stream=streams[k & 1].value)
    assert frame_of(0).tobytes() == h0.tobytes() and frame_of(1).tobytes() == h0.tobytes()
    for st, buf in zip(streams, bufs):
        ok(hip.hipStreamDestroy(st))
        ok(hip.hipFree(buf))
    # threads: each its own output, the same scene
    errs = []

    def worker(seed):
        try:
            for k in range(6):
                h, _ = sc.trace_primary(cam, W, H)
                if h.tobytes() != h0.tobytes():
                    errs.append((seed, k))
        except Exception as e:  # noqa: BLE001
            errs.append((seed, repr(e)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_frame_hints_api(pkg):
    for m in (-1, 0, 1, 2):
        pkg.set_frame_hints(m)
    pkg.set_frame_hints(-1)
    with pytest.raises(pkg.CgrtError):
        pkg.set_frame_hints(3)
