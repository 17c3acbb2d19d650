"""No GPU: the C-ABI library loads, exports every symbol include/cgrt.h declares, and reports errors the way
the header says (status codes + cgrt_last_error), including "no device => fail loudly, no CPU fallback"."""
import ctypes as C
import os
import re

import numpy as np
import pytest


def test_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(pkg.INCLUDE_DIR, "cgrt.h")).read()
    declared = sorted(set(re.findall(r"\b(cgrt_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 25
    L = C.CDLL(pkg.LIB_PATH)
    for sym in declared:
        assert hasattr(L, sym), f"libcgrt.so does not export {sym}"
    assert sorted(pkg.EXPORTS) == declared


def test_struct_layouts_match_reference_pods(pkg):
    assert pkg.RAY_DTYPE.itemsize == 28  # Ray, framework/include/ray.h:9-13
    assert pkg.HIT_DTYPE.itemsize == 16
    assert C.sizeof(pkg.Camera) == 36
    rs = pkg.record_sizes()
    assert rs["node"] == 64 and rs["tri"] == 64 and rs["hit"] == 16


def test_version_and_error_string(pkg):
    assert b"gfx950" in pkg.lib().cgrt_version()
    out = C.c_void_p()
    rc = pkg.lib().cgrt_scene_create(None, 3, None, None, 1, None, 1, None, 0, -1, C.byref(out))
    assert rc == -1 and b"NULL" in pkg.lib().cgrt_last_error()


def test_rejects_bad_indices(pkg, scene_data):
    sd = scene_data("cube")
    bad = pkg.scenes.SceneData(pos_nrm=sd.pos_nrm, tri=sd.tri.copy(), tri_mesh=sd.tri_mesh, materials=sd.materials)
    bad.tri[3, 1] = len(sd.pos_nrm)  # vertex out of range
    with pytest.raises(pkg.CgrtError) as e:
        pkg.Scene(bad, device=-1)
    assert e.value.code == -1
    bad = pkg.scenes.SceneData(pos_nrm=sd.pos_nrm, tri=sd.tri, tri_mesh=sd.tri_mesh[::-1].copy(), materials=sd.materials)
    with pytest.raises(pkg.CgrtError):
        pkg.Scene(bad, device=-1)


def test_host_only_scene_refuses_to_trace(pkg, scene_data):
    """There is no CPU traversal path: a scene without a device fails loudly."""
    s = pkg.Scene(scene_data("cube"), device=-1)
    rays = pkg.as_rays([[0, 0, -3]], [[0, 0, 1]])
    with pytest.raises(pkg.CgrtError) as e:
        s.intersect(rays)
    assert e.value.code == -2
    with pytest.raises(pkg.CgrtError) as e:
        s.trace_primary(pkg.scenes.default_camera(8, 8), 8, 8)
    assert e.value.code == -2


def test_no_device_is_an_error_not_a_fallback(pkg, scene_data):
    if pkg.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(pkg.CgrtError) as e:
        pkg.Scene(scene_data("cube"), device=0)
    assert e.value.code == -2
    with pytest.raises(pkg.CgrtError):
        pkg.ray_box(np.zeros((1, 6), np.float32), pkg.as_rays([[0, 0, 0]], [[0, 0, 1]]))


def test_product_does_not_reference_the_oracle(pkg):
    """The product tree must not import, link or name anything under oracle/."""
    root = os.path.dirname(pkg.LIB_PATH)
    pkgdir = os.path.dirname(root)
    for dp, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "cgrt_oracle" not in txt and "oracle/" not in txt, f"{f} mentions the oracle"


def test_option_setters_validate(pkg):
    L = pkg.lib()
    assert L.cgrt_set_primary_mode(2) == -1 and b"mode" in L.cgrt_last_error()
    assert L.cgrt_set_primary_mode(0) == 0
    assert L.cgrt_set_leaf_accel(1, 65) == -1
    assert L.cgrt_set_leaf_accel(1, 0) == 0


def test_render_rejects_bad_arguments(pkg, scene_data):
    s = pkg.Scene(scene_data("cube"), device=-1)
    import ctypes as C

    cam = pkg.Camera.from_array(pkg.scenes.default_camera(8, 8))
    rgb = np.zeros(8 * 8 * 3, np.float32)
    rc = pkg.lib().cgrt_render(s._h, C.byref(cam), 8, 8, None, 0, 2, rgb.ctypes.data_as(C.c_void_p), None)
    assert rc == -2  # host-only scene: no CPU shading/traversal path either
    rc = pkg.lib().cgrt_render(s._h, C.byref(cam), 8, 8, None, 3, 2, rgb.ctypes.data_as(C.c_void_p), None)
    assert rc == -1  # lights missing
