"""GPU tests of the boundary (include/cgrt.h) beyond single-call parity: the device brute force against the oracle's
(the F4 discriminator), one frame on several replicas from one caller (bytes equal the single-device frame), concurrent
per-ray calls on one handle as the reference's omp loop makes them, and loader/scene -> device render -> Screen -> BMP end
to end against bytes computed from the oracle's RGB."""
import os
import threading

import numpy as np
import pytest

import rayfam
from conftest import GOLDEN, bits, same_bits
from test_parity_gpu import _assert_hits_equal, _rays

pytestmark = pytest.mark.gpu
FMAX = rayfam.FMAX


# ---------------------------------------------------------------------------------------------------
# a8: intersectRayWithShape(Mesh) -- every triangle, no tree (ray_tracing.cpp:202-213)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["cube", "monkey", "cornell", "spheres"])
def test_brute_force_matches_oracle(pkg, orc, scene_data, name):
    sd = scene_data(name)
    o = orc.OracleScene(sd)
    _, boxes = o.nodes()
    W = H = 64
    fam = rayfam.families(sd, boxes, orc.generate_rays(pkg.scenes.default_camera(W, H), W, H), rng=np.random.RandomState(5), n_random=1500)
    rays = rayfam.concat(fam)
    sc = pkg.Scene(sd)
    hits, normals = sc.intersect_brute(_rays(pkg, rays))
    _assert_hits_equal(hits, normals, o.intersect(rays, brute_force=True), f"{name} brute force")


def test_brute_force_finds_what_the_reference_bvh_misses(pkg, orc, scene_data):
    """The three cube.obj rays of SURVEY.md F4: the reference's own brute-force loop returned t = 3.09774303 / 2.90451074 /
    2.88830972 (nine digits identify a binary32), its BVH misses.  Both outcomes on the device, bit for bit vs the oracle."""
    sd = scene_data("cube")
    sc, o = pkg.Scene(sd), orc.OracleScene(sd)
    rays = pkg.as_rays(np.broadcast_to(np.float32(rayfam.F4_ORIGIN), (3, 3)), np.float32(rayfam.F4_DIRS))
    hb, nb = sc.intersect_brute(rays)
    ref = o.intersect(rays, brute_force=True)
    _assert_hits_equal(hb, nb, ref, "F4 brute force")
    assert hb["hit"].tolist() == [1, 1, 1]
    assert [f"{t:.9g}" for t in hb["t"]] == rayfam.F4_BRUTE_T
    h, _ = sc.intersect(rays)
    assert h["hit"].tolist() == [0, 0, 0]


def test_brute_force_single_mesh(pkg, orc, scene_data):
    """mesh >= 0 is the reference's Mesh overload proper: one mesh, material never written, no spheres."""
    sd = scene_data("cornell")
    sc = pkg.Scene(sd)
    W = H = 48
    rays = orc.generate_rays(pkg.scenes.default_camera(W, H), W, H)
    for m in (0, 3, sd.nmesh - 1):
        sel = sd.tri_mesh == m
        sub = pkg.scenes.SceneData(pos_nrm=sd.pos_nrm, tri=sd.tri[sel], tri_mesh=np.zeros(int(sel.sum()), np.uint32), materials=sd.materials[m:m + 1])
        ref = orc.OracleScene(sub).intersect(rays, brute_force=True)
        hits, normals = sc.intersect_brute(_rays(pkg, rays), mesh=m)
        assert np.array_equal(hits["hit"], ref["hit"]) and same_bits(hits["t"], ref["t"]).all()
        assert (hits["material_id"] == -1).all()
        hm = ref["hit"] == 1
        assert np.array_equal(hits["prim_id"][hm], np.nonzero(sel)[0][ref["prim"][hm]])
        assert same_bits(normals[hm], ref["normal"][hm]).all()
    with pytest.raises(pkg.CgrtError):
        sc.intersect_brute(_rays(pkg, rays), mesh=sd.nmesh)


# ---------------------------------------------------------------------------------------------------
# one caller, N replicas, one framebuffer
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nrep", [2, 3])
def test_one_frame_on_several_replicas(pkg, orc, nrep):
    sd = pkg.scenes.make_dragon(40_000)
    W, H = 500, 301  # ragged right / bottom tiles
    cam = pkg.scenes.default_camera(W, H)
    reps = [pkg.Scene(sd, device=0) for _ in range(nrep)]  # devices {0, 0(, 0)}: replicas of the scene, as on N GPUs
    single, n_single = reps[0].trace_primary(cam, W, H, want_normals=True)
    hits, normals, st = pkg.trace_primary_multi(reps, cam, W, H, want_normals=True)
    assert hits.tobytes() == single.tobytes()
    m = single["hit"] == 1
    assert normals[m].tobytes() == n_single[m].tobytes() and not normals[~m].any()
    assert st["replicas"] == nrep and sum(st["rays"]) == W * H and min(st["rays"]) > 0.5 * W * H / nrep
    assert [r for r in st["rays"]] == [pkg.tiling.owned_pixels(W, H, r, nrep) for r in range(nrep)]
    ref = orc.OracleScene(sd).intersect(reps[0].generate_rays(cam, W, H))
    _assert_hits_equal(hits, None, ref, f"{nrep} replicas")
    with pytest.raises(pkg.CgrtError):
        pkg.trace_primary_multi([reps[0], reps[0]], cam, W, H)


@pytest.mark.parametrize("name,nrep,level", [("cornell", 2, 4), ("monkey", 3, 2)])
def test_one_shaded_frame_on_several_replicas(pkg, scene_data, name, nrep, level):
    sd = scene_data(name)
    W, H = 333, 217
    cam = pkg.scenes.default_camera(W, H)
    reps = [pkg.Scene(sd, device=0) for _ in range(nrep)]
    one, st1 = reps[0].render(cam, W, H, max_level=level)
    rgb, stn = pkg.render_multi(reps, cam, W, H, max_level=level)
    assert rgb.tobytes() == one.tobytes()
    for k in ("primary_rays", "shadow_rays", "reflection_rays"):
        assert stn[k] == st1[k]


# ---------------------------------------------------------------------------------------------------
# threads: the reference calls intersect on one const object from an omp parallel for (main.cpp:653-656)
# ---------------------------------------------------------------------------------------------------
def test_concurrent_per_ray_calls_on_one_handle(pkg, orc, scene_data):
    sd = scene_data("monkey")
    W = H = 64
    rays = orc.generate_rays(pkg.scenes.default_camera(W, H), W, H)
    bad, tim = pkg.host_threads_test(sd, rays, nthreads=8)  # 8 std::threads x BoundingVolumeHierarchy::intersect(Ray&, HitInfo&)
    assert bad == 0, f"{bad} disagreements between threaded per-ray calls and intersectBatch"
    assert tim["calls_per_second"] > 0


def test_concurrent_batches_from_python_threads(pkg, orc):
    """Eight threads, one Scene handle, batches of very different sizes interleaved with whole primary frames."""
    sd = pkg.scenes.make_dragon(30_000)
    sc = pkg.Scene(sd)
    W = H = 96
    cam = pkg.scenes.default_camera(W, H)
    rays = sc.generate_rays(cam, W, H)
    ref_h, ref_n = sc.intersect(rays)
    frame, _ = sc.trace_primary(cam, W, H)
    errors = []

    def worker(k):
        try:
            rng = np.random.RandomState(k)
            for it in range(30):
                n = int(rng.choice([1, 1, 2, 7, 64, 700, len(rays)]))
                i0 = rng.randint(0, len(rays) - n + 1)
                h, nn = sc.intersect(rays[i0:i0 + n])
                if h.tobytes() != ref_h[i0:i0 + n].tobytes() or nn.tobytes() != ref_n[i0:i0 + n].tobytes():
                    errors.append((k, it, "batch"))
                if it % 10 == 0:
                    f, _ = sc.trace_primary(cam, W, H)
                    if f.tobytes() != frame.tobytes():
                        errors.append((k, it, "frame"))
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]


# ---------------------------------------------------------------------------------------------------
# scene -> device render -> Screen -> BMP, end to end (screen.cpp:30-49)
# ---------------------------------------------------------------------------------------------------
def _bmp_from_rgb(rgb, W, H):
    """What Screen::setPixel + writeBitmapToFile must produce for an (H*W, 3) float frame indexed y*W+x with y UP:
    clamp to [0, 1], * 255 truncated to u8 (screen.cpp:38-49), image row 0 = top = largest y (setPixel's flip, :30-36),
    24-bit BMP: 54-byte header, rows bottom-up padded to 4 bytes, BGR."""
    img = np.clip(rgb.reshape(H, W, 3).astype(np.float32), 0.0, 1.0)
    u8 = (img * np.float32(255.0)).astype(np.uint8)  # float -> u8 truncation
    row_bytes = (W * 3 + 3) & ~3
    data = np.zeros((H, row_bytes), np.uint8)
    # Screen row r (top-down) = frame row y = H-1-r; BMP stores Screen rows bottom-up, i.e. frame rows y = 0, 1, ...
    data[:, :W * 3] = u8[:, :, ::-1].reshape(H, W * 3)
    hdr = bytearray(54)
    hdr[0:2] = b"BM"
    hdr[2:6] = (54 + row_bytes * H).to_bytes(4, "little")
    hdr[10:14] = (54).to_bytes(4, "little")
    hdr[14:18] = (40).to_bytes(4, "little")
    hdr[18:22] = W.to_bytes(4, "little")
    hdr[22:26] = H.to_bytes(4, "little")
    hdr[26:28] = (1).to_bytes(2, "little")
    hdr[28:30] = (24).to_bytes(2, "little")
    hdr[34:38] = (row_bytes * H).to_bytes(4, "little")
    return bytes(hdr) + data.tobytes()


@pytest.mark.parametrize("nrep", [1, 2])
def test_end_to_end_image(pkg, orc, scene_data, tmp_path, nrep):
    """Cornell 480x270, depth 2, rendered on the device through the C++ mirror (renderRayTracingOnDevices -> Screen ->
    writeBitmapToFile): the file's bytes equal the BMP computed here from the ORACLE's RGB wherever the 8-bit value does not
    sit within the 1e-5 RGB tolerance of a truncation boundary, and from the device's own float frame everywhere."""
    sd = scene_data("cornell")
    W, H = 480, 270
    cam = pkg.scenes.default_camera(W, H)
    path = str(tmp_path / "render.bmp")
    rgb = pkg.host_render_bmp(sd, cam, W, H, path, max_level=2, nreplicas=nrep)
    got = open(path, "rb").read()
    assert got == _bmp_from_rgb(rgb, W, H), "Screen/BMP bytes differ from the device's float frame"
    ref, _ = orc.OracleScene(sd).render(cam, W, H, sd.point_lights, max_level=2)
    assert np.abs(rgb.astype(np.float64) - ref).max() <= 1e-5  # north_star tolerance on RGB
    want = _bmp_from_rgb(ref.astype(np.float32), W, H)
    a = np.frombuffer(got, np.uint8)[54:].astype(np.int16)
    b = np.frombuffer(want, np.uint8)[54:].astype(np.int16)
    assert got[:54] == want[:54]
    diff = np.nonzero(a != b)[0]
    # a differing byte must come from a channel value within 1e-5 * 255 of an integer (truncation boundary)
    v = np.clip(ref.astype(np.float64), 0, 1) * 255.0
    near = np.abs(v - np.round(v)) <= 1e-5 * 255.0 + 1e-9
    assert len(diff) <= near.sum() and np.abs(a - b).max(initial=0) <= 1
    assert (ref.sum(1) > 0).mean() > 0.05


# ---------------------------------------------------------------------------------------------------
# call combining: concurrent per-ray calls share launches inside the library (capi.cpp combined_intersect)
# ---------------------------------------------------------------------------------------------------
def test_64_threads_of_per_ray_calls_equal_one_batch(pkg, orc):
    """64 threads x 10 000 BoundingVolumeHierarchy::intersect(Ray&, HitInfo&) calls on ONE object (the
    reference's omp parallel for, main.cpp:653-656, scaled up) must return what one batch of the same 640 000 rays returns:
    ray.t bits, the flag, hitInfo.normal bits and the material, ray for ray."""
    sd = pkg.scenes.make_dragon(60_000)
    W = H = 800  # 640 000 primary rays: 10 000 per thread
    rays = orc.generate_rays(pkg.scenes.default_camera(W, H), W, H)
    bad, tim = pkg.host_threads_test(sd, rays, nthreads=64)
    assert bad == 0, f"{bad} disagreements between combined per-ray calls and intersectBatch"
    print(f"64 threads: {tim['calls_per_second']:.0f} per-ray calls/s, one thread {tim['us_per_call_one_thread']:.1f} us per call")
    assert tim["calls_per_second"] > 100_000  # (round 2: 84 K/s from 8 threads, one launch per ray)


def test_call_combining_on_off_and_mixed_sizes(pkg, orc, scene_data):
    """Small calls of different sizes (1 .. 64 rays, with and without a normals array) from 16 threads, combining on and off: the
    same bytes as one batch.  HitInfo stays untouched on a miss through the combined path too."""
    import threading

    sd = scene_data("monkey")
    sc = pkg.Scene(sd)
    W = H = 96
    rays = sc.generate_rays(pkg.scenes.default_camera(W, H), W, H)
    ref_h, ref_n = sc.intersect(rays)  # one batch (9 216 rays: the direct path)
    assert 0 < (ref_h["hit"] == 1).sum() < len(rays)
    for combining in (True, False, True):
        pkg.set_call_combining(combining)
        errors = []

        def worker(k):
            try:
                rng = np.random.RandomState(100 + k)
                for it in range(120):
                    n = int(rng.choice([1, 1, 1, 2, 5, 33, 64]))
                    i0 = rng.randint(0, len(rays) - n + 1)
                    want_n = bool(rng.randint(0, 2))
                    h, nn = sc.intersect(rays[i0:i0 + n], want_normals=want_n)
                    if h.tobytes() != ref_h[i0:i0 + n].tobytes():
                        errors.append(("hits", k, it, n))
                    if want_n and nn.tobytes() != ref_n[i0:i0 + n].tobytes():  # zeros where the ray missed: untouched
                        errors.append(("normals", k, it, n))
            except Exception as ex:  # noqa: BLE001
                errors.append(repr(ex))

        th = [threading.Thread(target=worker, args=(k,)) for k in range(16)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errors, errors[:5]
    pkg.set_call_combining(True)


@pytest.mark.parametrize("name,W,H,level,threads", [("monkey", 160, 120, 2, 32), ("cornell", 120, 90, 3, 16), ("cube", 64, 64, 2, 1)])
def test_per_ray_driver_matches_oracle(pkg, orc, scene_data, name, W, H, level, threads):
    """The reference's driver taken literally through the mirror (renderToBufferPerRay: omp parallel for over rows, per-pixel
    recursion, ONE BoundingVolumeHierarchy::intersect call per ray): RGB within 1e-5 of the oracle's recursive driver, the same
    rays cast, and the same frame as the batched wavefront of the mirror."""
    sd = scene_data(name)
    cam = pkg.scenes.default_camera(W, H)
    rgb, st = pkg.host_render_per_ray(sd, cam, W, H, level, threads=threads)
    ref, nrays = orc.OracleScene(sd).render(cam, W, H, sd.point_lights, max_level=level)
    err = np.abs(rgb.astype(np.float64) - ref).max()
    assert err <= 1e-5, f"max abs RGB error {err}"
    assert st["primary"] + st["shadow"] + st["reflection"] == nrays
    wave, _ = pkg.host_render(sd, cam, W, H, level)
    assert np.abs(rgb - wave).max() <= 1e-6


@pytest.mark.gpu
def test_large_host_lists_go_through_the_bounce_buffers_unchanged(pkg, orc):
    """Transfers above 1 MB take another road than small ones (capi.cpp lane_upload / lane_download: two alternating pinned 8 MB
    buffers, pieces copied on several threads, and of a long list only the normals of rays that hit are copied back): a list of
    700 K rays -- several pieces per buffer, a ragged last piece -- must come back exactly as its 20 K-ray slices do, HitInfo of a
    miss untouched (the caller's normal stays), through the tree and through the brute force; and a whole 4 M-pixel frame's hits
    and normals equal the same frame fetched in rectangles."""
    sd = pkg.scenes.make_dragon(30_000)
    sc = pkg.Scene(sd)
    W, H = 1100, 640  # 704 000 rays: 22.5 MB of rays, 11.3 MB of hits, 8.4 MB of normals
    cam = pkg.scenes.default_camera(W, H)
    rays = sc.generate_rays(cam, W, H)
    assert len(rays) * 28 > 2 * (8 << 20)
    sentinel = np.float32(7.25)

    def run(fn, sl):
        hits = np.zeros(sl.stop - sl.start, pkg.HIT_DTYPE)
        normals = np.full((sl.stop - sl.start, 3), sentinel, np.float32)
        fn(rays[sl], hits, normals)
        return hits, normals

    def tree(r, hits, normals):
        pkg._check(pkg.lib().cgrt_intersect_batch(sc._h, pkg._ptr(r), len(r), pkg._ptr(hits), pkg._ptr(normals)))

    def brute(r, hits, normals):
        pkg._check(pkg.lib().cgrt_intersect_brute_batch(sc._h, pkg._ptr(r), len(r), -1, pkg._ptr(hits), pkg._ptr(normals)))

    whole_h, whole_n = run(tree, slice(0, len(rays)))
    step = 20_000
    for a in range(0, len(rays), step):
        b = min(len(rays), a + step)
        h, n = run(tree, slice(a, b))
        assert h.tobytes() == whole_h[a:b].tobytes() and n.tobytes() == whole_n[a:b].tobytes(), a
    miss = whole_h["hit"] == 0
    assert miss.any() and (~miss).any()
    assert (whole_n[miss] == sentinel).all() and not (whole_n[~miss] == sentinel).all(axis=1).any()
    # against the oracle on a sample (the whole list would take the CPU a while)
    pick = np.random.RandomState(3).choice(len(rays), 20_000, replace=False)
    _assert_hits_equal(whole_h[pick], whole_n[pick], orc.OracleScene(sd).intersect(rays[pick]), "700 K-ray list through the bounce buffers")
    # the brute force takes the same road (a shorter list: it tests every triangle)
    nb = 90_000
    bh, bn = run(brute, slice(0, nb))
    for a in range(0, nb, 30_000):
        h, n = run(brute, slice(a, a + 30_000))
        assert h.tobytes() == bh[a : a + 30_000].tobytes() and n.tobytes() == bn[a : a + 30_000].tobytes(), a
    # a frame: the whole-frame road (normals by hit flag) against rectangles (seeded on the device)
    Wf, Hf = 2048, 2048
    camf = pkg.scenes.default_camera(Wf, Hf)
    fh, fn_ = sc.trace_primary(camf, Wf, Hf, want_normals=True)
    for rect in ((0, 0, 1024, 1024), (1024, 0, 2048, 1024), (0, 1024, 2048, 2048)):
        rh, rn = sc.trace_primary(camf, Wf, Hf, rect=rect, want_normals=True)
        x0, y0, x1, y1 = rect
        m = np.zeros((Hf, Wf), bool)
        m[y0:y1, x0:x1] = True
        m = m.ravel()
        assert rh[m].tobytes() == fh[m].tobytes() and rn[m].tobytes() == fn_[m].tobytes(), rect
