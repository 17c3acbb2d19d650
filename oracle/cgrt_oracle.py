"""ctypes binding of the CPU oracle (oracle/cgrt_oracle.cpp).

TEST INFRASTRUCTURE ONLY -- see the header of cgrt_oracle.cpp.  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_build", "libcgrt_oracle.so")
LIB_O0 = os.path.join(_HERE, "_build", "libcgrt_oracle_O0.so")

OHIT = np.dtype(
    [("t", np.float32), ("prim", np.uint32), ("material", np.int32), ("hit", np.uint32), ("normal", np.float32, 3), ("pad", np.float32)]
)
assert OHIT.itemsize == 32


def build(verbose: bool = False) -> None:
    r = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout[-3000:], r.stderr[-3000:])
    if r.returncode:
        raise RuntimeError("building the oracle failed")


_libs = {}


def lib(o0: bool = False) -> C.CDLL:
    path = LIB_O0 if o0 else LIB
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    L.oracle_scene_create.argtypes = [vp, u32, vp, vp, u32, vp, u32, vp, u32]
    L.oracle_scene_create.restype = vp
    L.oracle_scene_destroy.argtypes = [vp]
    L.oracle_scene_destroy.restype = None
    L.oracle_num_levels.argtypes = [vp]
    L.oracle_num_nodes.argtypes = [vp]
    L.oracle_build_seconds.argtypes = [vp]
    L.oracle_build_seconds.restype = C.c_double
    L.oracle_get_nodes.argtypes = [vp, vp, vp]
    L.oracle_get_nodes.restype = None
    L.oracle_leaf_prims.argtypes = [vp, i32, vp, u32]
    L.oracle_leaf_prims.restype = u32
    L.oracle_intersect_batch.argtypes = [vp, vp, u64, vp, vp, i32, i32]
    L.oracle_intersect_batch.restype = None
    L.oracle_generate_rays.argtypes = [vp] + [i32] * 6 + [vp]
    L.oracle_generate_rays.restype = None
    L.oracle_trace_primary_timed.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32]
    L.oracle_trace_primary_timed.restype = C.c_double
    L.oracle_render.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32, i32, vp, i32]
    L.oracle_render.restype = u64
    L.oracle_render_soft.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32, vp, i32, vp, C.c_uint32, C.c_uint32, C.c_uint32, i32, vp, i32]
    L.oracle_render_soft.restype = u64
    for name in ("oracle_ray_triangle", "oracle_ray_plane", "oracle_ray_box", "oracle_ray_sphere"):
        f = getattr(L, name)
        f.argtypes = [vp, vp, u64, vp]
        f.restype = None
    L.oracle_triangle_plane.argtypes = [vp, u64, vp]
    L.oracle_triangle_plane.restype = None
    L.oracle_point_in_triangle.argtypes = [vp, u64, vp]
    L.oracle_point_in_triangle.restype = None
    L.oracle_max_threads.restype = i32
    _libs[path] = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a, shape):
    return np.ascontiguousarray(np.asarray(a, np.float32).reshape(shape))


def rays7(rays) -> np.ndarray:
    """Accepts the product's structured RAY dtype or an (n,7) float array."""
    a = np.ascontiguousarray(rays)
    if a.dtype.fields is not None:
        a = a.view(np.float32).reshape(-1, 7)
    return _f32(a, (-1, 7))


class OracleScene:
    def __init__(self, sd, o0: bool = False):
        self.L = lib(o0)
        pn = _f32(sd.pos_nrm, (-1, 6))
        tri = np.ascontiguousarray(sd.tri, np.uint32).reshape(-1, 3)
        tm = np.ascontiguousarray(sd.tri_mesh, np.uint32)
        mats = _f32(sd.materials, (-1, 8))
        sph = _f32(sd.spheres, (-1, 5))
        self.h = self.L.oracle_scene_create(_p(pn), len(pn), _p(tri), _p(tm), len(tri), _p(mats), len(mats), _p(sph), len(sph))
        if not self.h:
            raise ValueError("oracle rejected the scene arrays")
        self.ntris = len(tri)

    def close(self):
        if getattr(self, "h", None):
            self.L.oracle_scene_destroy(self.h)
            self.h = None

    __del__ = close

    def num_levels(self) -> int:
        return int(self.L.oracle_num_levels(self.h))

    def build_seconds(self) -> float:
        return float(self.L.oracle_build_seconds(self.h))

    def nodes(self):
        n = int(self.L.oracle_num_nodes(self.h))
        meta = np.zeros((n, 5), np.int32)
        boxes = np.zeros((n, 6), np.float32)
        if n:
            self.L.oracle_get_nodes(self.h, _p(meta), _p(boxes))
        return meta, boxes

    def leaf_prims(self, node: int) -> np.ndarray:
        n = int(self.L.oracle_leaf_prims(self.h, node, None, 0))
        out = np.zeros(n, np.uint32)
        if n:
            self.L.oracle_leaf_prims(self.h, node, _p(out), n)
        return out

    def intersect(self, rays, brute_force: bool = False, counters: bool = False, threads: int = 0):
        r = rays7(rays)
        out = np.zeros(len(r), OHIT)
        cnt = np.zeros(4, np.uint64) if counters else None
        self.L.oracle_intersect_batch(self.h, _p(r), len(r), _p(out), _p(cnt), 1 if brute_force else 0, threads)
        if counters:
            return out, dict(zip(("inner_visits", "leaf_visits", "tri_tests", "box_tests"), (int(x) for x in cnt)))
        return out

    def trace_primary_timed(self, cam, W, H, y0=0, y1=None, want_hits=False, threads=0):
        y1 = H if y1 is None else y1
        cam = _f32(cam, (9,))
        out = np.zeros((y1 - y0) * W, OHIT) if want_hits else None
        sec = float(self.L.oracle_trace_primary_timed(self.h, _p(cam), W, H, y0, y1, _p(out), threads))
        return sec, out

    def render(self, cam, W, H, lights, max_level=2, y0=0, y1=None, threads=0):
        y1 = H if y1 is None else y1
        cam = _f32(cam, (9,))
        lights = _f32(lights, (-1, 6))
        rgb = np.zeros(((y1 - y0) * W, 3), np.float32)
        n = int(self.L.oracle_render(self.h, _p(cam), W, H, y0, y1, _p(lights), len(lights), max_level, _p(rgb), threads))
        return rgb, n


    def render_soft(self, cam, W, H, lights, spherical, units, samples=200, seed=0, max_level=2, y0=0, y1=None, threads=0):
        """main.cpp:168-218 with the randomUnitVector() draws taken from `units` (n x 3), see cgrt_oracle.cpp Shader."""
        y1 = H if y1 is None else y1
        cam = _f32(cam, (9,))
        lights = _f32(lights, (-1, 6))
        spherical = _f32(spherical, (-1, 7))
        units = _f32(units, (-1, 3))
        rgb = np.zeros(((y1 - y0) * W, 3), np.float32)
        n = int(self.L.oracle_render_soft(self.h, _p(cam), W, H, y0, y1, _p(lights), len(lights), _p(spherical), len(spherical),
                                          _p(units), len(units), samples, seed, max_level, _p(rgb), threads))
        return rgb, n


def generate_rays(cam, W, H, rect=None) -> np.ndarray:
    x0, y0, x1, y1 = rect if rect is not None else (0, 0, W, H)
    cam = _f32(cam, (9,))
    out = np.zeros(((x1 - x0) * (y1 - y0), 7), np.float32)
    lib().oracle_generate_rays(_p(cam), W, H, x0, y0, x1, y1, _p(out))
    return out


def ray_triangle(tri18, rays):
    a, r = _f32(tri18, (-1, 18)), rays7(rays)
    out = np.zeros(len(r), OHIT)
    lib().oracle_ray_triangle(_p(a), _p(r), len(r), _p(out))
    return out


def ray_plane(plane4, rays):
    a, r = _f32(plane4, (-1, 4)), rays7(rays)
    out = np.zeros(len(r), OHIT)
    lib().oracle_ray_plane(_p(a), _p(r), len(r), _p(out))
    return out


def ray_box(box6, rays):
    a, r = _f32(box6, (-1, 6)), rays7(rays)
    out = np.zeros(len(r), OHIT)
    lib().oracle_ray_box(_p(a), _p(r), len(r), _p(out))
    return out


def ray_sphere(sph4, rays):
    a, r = _f32(sph4, (-1, 4)), rays7(rays)
    out = np.zeros(len(r), OHIT)
    lib().oracle_ray_sphere(_p(a), _p(r), len(r), _p(out))
    return out


def triangle_plane(tri9):
    a = _f32(tri9, (-1, 9))
    out = np.zeros((len(a), 4), np.float32)
    lib().oracle_triangle_plane(_p(a), len(a), _p(out))
    return out


def point_in_triangle(in15):
    a = _f32(in15, (-1, 15))
    out = np.zeros(len(a), np.uint8)
    lib().oracle_point_in_triangle(_p(a), len(a), _p(out))
    return out


def max_threads() -> int:
    return int(lib().oracle_max_threads())
