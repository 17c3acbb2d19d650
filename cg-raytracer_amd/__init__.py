"""cg-raytracer_amd: MI355X-native BVH traversal + ray/triangle intersection (the hot path of
mgokbulut/CG-RayTracer) behind a C-ABI (include/cgrt.h).

This module is the thin Python host used by tests/ and bench.py: a ctypes binding of ``libcgrt.so``
(built in-tree by ``build_native()`` / ``__graft_entry__.build()``).  There is no CPU fallback: if the
library is missing, or no HIP device is usable, calls raise.

The directory name contains a hyphen, so it is loaded under the module name ``cg_raytracer_amd``
(see ``__graft_entry__.load_package``).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

from . import scenes, tiling  # noqa: F401  (re-export)
from .scenes import SceneData

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", os.environ.get("CGRT_LIB_NAME", "libcgrt.so"))  # CGRT_LIB_NAME: experiment builds
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")

NO_PRIM = 0xFFFFFFFF

RAY_DTYPE = np.dtype([("origin", np.float32, 3), ("direction", np.float32, 3), ("t", np.float32)])
HIT_DTYPE = np.dtype([("t", np.float32), ("prim_id", np.uint32), ("material_id", np.int32), ("hit", np.uint32)])
assert RAY_DTYPE.itemsize == 28 and HIT_DTYPE.itemsize == 16


class CgrtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"cgrt error {code}: {msg}")
        self.code = code


class Counters(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64),
        ("inner_visits", C.c_uint64),
        ("leaf_visits", C.c_uint64),
        ("tri_tests", C.c_uint64),
        ("sub_visits", C.c_uint64),
        ("cert_boxes", C.c_uint64),
        ("fallback_rays", C.c_uint64),
        ("tree_rays", C.c_uint64),
    ]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class RenderStats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("reflection_rays", C.c_uint64), ("levels", C.c_int32),
                ("device_ms", C.c_float), ("soft_shadow_rays", C.c_uint64)]


class MultiStats(C.Structure):
    _fields_ = [("replicas", C.c_int32), ("kernel_ms_max", C.c_float), ("download_ms_max", C.c_float), ("wall_ms", C.c_double),
                ("rays", C.c_uint64 * 64)]


class SoftShadows(C.Structure):
    _fields_ = [("spherical", C.c_void_p), ("unit_vectors", C.c_void_p), ("nspherical", C.c_uint32), ("samples", C.c_uint32),
                ("nunits", C.c_uint32), ("seed", C.c_uint32), ("closest_hit", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [
        ("look_at", C.c_float * 3),
        ("euler", C.c_float * 3),
        ("distance", C.c_float),
        ("fovy", C.c_float),
        ("aspect", C.c_float),
    ]

    @staticmethod
    def from_array(a) -> "Camera":
        a = np.asarray(a, np.float32)
        cam = Camera()
        cam.look_at[:] = a[0:3]
        cam.euler[:] = a[3:6]
        cam.distance, cam.fovy, cam.aspect = float(a[6]), float(a[7]), float(a[8])
        return cam


HOST_LIB_PATH = os.path.join(_HERE, "lib", "libcgrt_host.so")


def build_native(verbose: bool = False) -> str:
    """Compile libcgrt.so for gfx950 in-tree (hipcc cross-compiles without a GPU), then the C++ host mirror of the
    reference interface (libcgrt_host.so + the headless `render` tool, g++)."""
    for sub in ("csrc", "host"):
        r = subprocess.run(["make", "-C", os.path.join(_HERE, sub)], capture_output=True, text=True)
        if verbose or r.returncode:
            print(r.stdout[-4000:], r.stderr[-4000:])
        if r.returncode:
            raise RuntimeError(f"building {sub} failed")
    return LIB_PATH


_lib: Optional[C.CDLL] = None

# every symbol include/cgrt.h declares
EXPORTS = [
    "cgrt_scene_create", "cgrt_scene_destroy", "cgrt_set_leaf_accel", "cgrt_num_subnodes", "cgrt_set_primary_mode", "cgrt_set_kernel_shape", "cgrt_get_kernel_shape", "cgrt_set_render_prediction", "cgrt_set_frame_hints", "cgrt_debug_set_hint_thresholds", "cgrt_debug_hint_counts", "cgrt_debug_render_path", "cgrt_set_fast_tree", "cgrt_scene_set_walk", "cgrt_scene_walk", "cgrt_scene_build_info", "cgrt_num_levels", "cgrt_num_nodes", "cgrt_get_nodes", "cgrt_leaf_prims",
    "cgrt_build_seconds", "cgrt_device_bytes", "cgrt_intersect_batch", "cgrt_set_call_combining", "cgrt_debug_combiner_stats", "cgrt_intersect_brute_batch", "cgrt_intersect_batch_device", "cgrt_trace_primary",
    "cgrt_trace_primary_device", "cgrt_generate_rays", "cgrt_render", "cgrt_render_soft", "cgrt_render_mapped", "cgrt_render_rank", "cgrt_render_counted", "cgrt_trace_primary_multi", "cgrt_render_multi", "cgrt_count_primary", "cgrt_count_batch", "cgrt_debug_wave_times", "cgrt_debug_fastdiv_check", "cgrt_debug_gather_calibration", "cgrt_debug_check_layout", "cgrt_debug_layout_hash", "cgrt_set_build_threads", "cgrt_record_sizes",
    "cgrt_ray_triangle_batch", "cgrt_ray_plane_batch", "cgrt_ray_box_batch", "cgrt_ray_sphere_batch",
    "cgrt_triangle_plane_batch", "cgrt_point_in_triangle_batch", "cgrt_device_count", "cgrt_last_error", "cgrt_version", "cgrt_source_hash",
]  # fmt: skip


def lib() -> C.CDLL:
    """Load libcgrt.so; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        try:  # not shipped with this checkout: compile it now (hipcc, gfx950); there is no other path to fall back to
            build_native()
        except Exception as e:  # noqa: BLE001
            raise RuntimeError(f"{LIB_PATH} is missing and could not be built: {e}") from e
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    L.cgrt_last_error.restype = C.c_char_p
    L.cgrt_version.restype = C.c_char_p
    L.cgrt_source_hash.restype = C.c_char_p
    L.cgrt_scene_create.argtypes = [vp, u32, vp, vp, u32, vp, u32, vp, u32, i32, C.POINTER(vp)]
    L.cgrt_scene_destroy.argtypes = [vp]
    L.cgrt_scene_destroy.restype = None
    for f in (L.cgrt_num_levels, L.cgrt_num_nodes):
        f.argtypes = [vp]
    L.cgrt_set_leaf_accel.argtypes = [i32, i32]
    L.cgrt_num_subnodes.argtypes = [vp]
    L.cgrt_debug_check_layout.argtypes = [vp]
    L.cgrt_debug_layout_hash.argtypes = [vp, C.POINTER(u64)]
    L.cgrt_set_build_threads.argtypes = [i32]
    L.cgrt_set_primary_mode.argtypes = [i32]
    L.cgrt_set_fast_tree.argtypes = [i32]
    L.cgrt_set_kernel_shape.argtypes = [i32, u64]
    L.cgrt_set_render_prediction.argtypes = [i32]
    L.cgrt_set_frame_hints.argtypes = [i32]
    L.cgrt_debug_set_hint_thresholds.argtypes = [u32, u32]
    L.cgrt_debug_render_path.argtypes = [C.c_void_p]
    L.cgrt_get_kernel_shape.argtypes = [C.POINTER(i32), C.POINTER(u64)]
    L.cgrt_scene_set_walk.argtypes = [vp, i32]
    L.cgrt_scene_walk.argtypes = [vp]
    L.cgrt_scene_build_info.argtypes = [vp, vp]
    L.cgrt_get_nodes.argtypes = [vp, vp, vp]
    L.cgrt_leaf_prims.argtypes = [vp, i32, vp, u32]
    L.cgrt_leaf_prims.restype = C.c_int64
    L.cgrt_build_seconds.argtypes = [vp]
    L.cgrt_build_seconds.restype = C.c_double
    L.cgrt_device_bytes.argtypes = [vp]
    L.cgrt_device_bytes.restype = u64
    L.cgrt_intersect_batch.argtypes = [vp, vp, u64, vp, vp]
    L.cgrt_intersect_brute_batch.argtypes = [vp, vp, u64, i32, vp, vp]
    L.cgrt_set_call_combining.argtypes = [i32]
    L.cgrt_debug_combiner_stats.argtypes = [vp, vp]
    L.cgrt_intersect_batch_device.argtypes = [vp, vp, u64, vp, vp, vp]
    L.cgrt_trace_primary.argtypes = [vp, C.POINTER(Camera)] + [i32] * 8 + [vp, vp]
    L.cgrt_trace_primary_device.argtypes = [vp, C.POINTER(Camera)] + [i32] * 8 + [vp, vp, vp]
    L.cgrt_generate_rays.argtypes = [vp, C.POINTER(Camera)] + [i32] * 6 + [vp]
    L.cgrt_render.argtypes = [vp, C.POINTER(Camera), i32, i32, vp, u32, i32, vp, C.POINTER(RenderStats)]
    L.cgrt_render_counted.argtypes = [vp, C.POINTER(Camera), i32, i32, vp, u32, i32, vp, C.POINTER(RenderStats), C.POINTER(Counters)]
    L.cgrt_render_soft.argtypes = [vp, C.POINTER(Camera), i32, i32, vp, u32, C.POINTER(SoftShadows), i32, vp, C.POINTER(RenderStats)]
    L.cgrt_render_mapped.argtypes = [vp, C.POINTER(Camera), i32, i32, vp, u32, C.POINTER(SoftShadows), i32, C.POINTER(vp), C.POINTER(RenderStats)]
    L.cgrt_render_rank.argtypes = [vp, C.POINTER(Camera), i32, i32, vp, u32, C.POINTER(SoftShadows), i32, i32, i32, vp, C.POINTER(RenderStats)]
    L.cgrt_trace_primary_multi.argtypes = [C.POINTER(vp), i32, C.POINTER(Camera), i32, i32, vp, vp, C.POINTER(MultiStats)]
    L.cgrt_render_multi.argtypes = [C.POINTER(vp), i32, C.POINTER(Camera), i32, i32, vp, u32, C.POINTER(SoftShadows), i32, vp, C.POINTER(RenderStats)]
    L.cgrt_count_primary.argtypes = [vp, C.POINTER(Camera)] + [i32] * 8 + [C.POINTER(Counters)]
    L.cgrt_count_batch.argtypes = [vp, vp, u64, C.POINTER(Counters)]
    L.cgrt_debug_gather_calibration.argtypes = [i32, u64, i32]
    L.cgrt_debug_fastdiv_check.argtypes = [i32, vp, vp, u64, vp, vp]
    L.cgrt_debug_wave_times.argtypes = [vp, C.POINTER(Camera), i32, i32, vp, u64]
    L.cgrt_record_sizes.argtypes = [C.POINTER(u32)] * 4
    L.cgrt_record_sizes.restype = None
    L.cgrt_ray_triangle_batch.argtypes = [i32, vp, vp, u64, vp, vp, vp]
    L.cgrt_ray_plane_batch.argtypes = [i32, vp, vp, u64, vp, vp]
    L.cgrt_ray_box_batch.argtypes = [i32, vp, vp, u64, vp, vp, vp]
    L.cgrt_ray_sphere_batch.argtypes = [i32, vp, vp, u64, vp, vp, vp]
    L.cgrt_triangle_plane_batch.argtypes = [i32, vp, u64, vp]
    L.cgrt_point_in_triangle_batch.argtypes = [i32, vp, u64, vp]
    _lib = L
    return L


def _check(rc: int) -> None:
    if rc != 0:
        raise CgrtError(rc, lib().cgrt_last_error().decode())


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a, shape) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, np.float32).reshape(shape))


def as_rays(origin, direction, t=None) -> np.ndarray:
    o = np.asarray(origin, np.float32).reshape(-1, 3)
    d = np.asarray(direction, np.float32).reshape(-1, 3)
    r = np.zeros(len(o), RAY_DTYPE)
    r["origin"], r["direction"] = o, d
    r["t"] = np.float32(np.finfo(np.float32).max) if t is None else np.asarray(t, np.float32)
    return r


def _as_ray_array(rays) -> np.ndarray:
    """RAY_DTYPE array from a RAY_DTYPE array or an (n, 7) float32 array {origin, direction, t} (viewed, never cast
    element by element)."""
    a = np.asarray(rays)
    if a.dtype != RAY_DTYPE:
        a = np.ascontiguousarray(a, np.float32)
        if a.ndim != 2 or a.shape[1] != 7:
            raise ValueError("rays must be a RAY_DTYPE array or an (n, 7) float32 array")
        a = a.view(RAY_DTYPE).reshape(-1)
    return np.ascontiguousarray(a)


def record_sizes() -> dict:
    v = [C.c_uint32() for _ in range(4)]
    lib().cgrt_record_sizes(*[C.byref(x) for x in v])
    return dict(zip(("node", "tri", "sub", "hit"), (int(x.value) for x in v)))


def set_leaf_accel(enabled: bool = True, sub_leaf_tris: int = 0) -> None:
    """Process-wide build option for scenes created afterwards (results are identical either way)."""
    _check(lib().cgrt_set_leaf_accel(1 if enabled else 0, sub_leaf_tris))


def set_primary_mode(mode: int) -> None:
    """0 = one wave per 8x8 tile, 1 = persistent waves with lane refill (same results)."""
    _check(lib().cgrt_set_primary_mode(int(mode)))


def set_kernel_shape(mode: int = -1, max_rays: int = 0) -> None:
    """-1 = by launch size (short lists are laid out sparsely: include/cgrt.h), 0 = 64 rays per wave always, 1 = quad per ray
    (16 per wave, frames too), 2 = 16 rays per wave for every list, 3 = 4 rays per wave for every list; max_rays = 0 keeps the
    threshold.  Same results whatever the shape."""
    _check(lib().cgrt_set_kernel_shape(int(mode), int(max_rays)))


def set_frame_hints(mode: int = -1) -> None:
    """Primary frames: -1 by frame size, 0 off, 1 the previous frame's hard tiles first, 2 hard tiles as four 16-ray waves; same pixels."""
    _check(lib().cgrt_set_frame_hints(int(mode)))


def debug_set_hint_thresholds(dense_ticks: int = 0, sparse_ticks: int = 0) -> None:
    _check(lib().cgrt_debug_set_hint_thresholds(int(dense_ticks), int(sparse_ticks)))


def set_render_prediction(enabled: bool = True) -> None:
    """cgrt_render*: size a frame's launches from the scene's previous frame of the same shape (default) or wait for the device's
    counts inside every frame; same pixels."""
    _check(lib().cgrt_set_render_prediction(1 if enabled else 0))


def kernel_shape() -> Tuple[int, int]:
    m, r = C.c_int(), C.c_uint64()
    _check(lib().cgrt_get_kernel_shape(C.byref(m), C.byref(r)))
    return int(m.value), int(r.value)


def set_call_combining(enabled: bool = True) -> None:
    """Small cgrt_intersect_batch calls of concurrent threads share one launch (default) or launch one by one; same results."""
    _check(lib().cgrt_set_call_combining(1 if enabled else 0))


def set_build_threads(threads: int = 0) -> None:
    """Worker threads of the host builders for scenes created afterwards (0 = hardware concurrency); the arrays do not depend on it."""
    _check(lib().cgrt_set_build_threads(int(threads)))


def set_fast_tree(mode: int) -> None:
    """Process-wide build option: -1 = fast tree for scenes with fat leaves (default), 0 = never, 1 = whenever possible."""
    _check(lib().cgrt_set_fast_tree(int(mode)))


def source_hash() -> str:
    """sha256[:16] of the sources the loaded library was built from (include/cgrt.h cgrt_source_hash)."""
    return lib().cgrt_source_hash().decode()


def device_count() -> int:
    return int(lib().cgrt_device_count())


def unit_vector_table(n: int = 1 << 16, seed: int = 0) -> np.ndarray:
    """n draws of the reference's randomUnitVector() (main.cpp:46-59: three N(0,1) floats, glm::normalize) as input data
    for render_soft.  Any generator will do -- the table is data, not part of the parity contract."""
    g = np.random.default_rng(seed).standard_normal((n, 3)).astype(np.float32)
    d = (g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1]) + g[:, 2] * g[:, 2]
    inv = np.float32(1.0) / np.sqrt(d, dtype=np.float32)
    return np.ascontiguousarray(g * inv[:, None], dtype=np.float32)


class Scene:
    """Owns a CgrtScene: the host mirror of ``BoundingVolumeHierarchy(Scene*)`` (bvh.h:39-48)."""

    def __init__(self, sd: SceneData, device: int = 0):
        self.sd = sd
        self._h = C.c_void_p()
        pn = _f32(sd.pos_nrm, (-1, 6))
        tri = np.ascontiguousarray(sd.tri, np.uint32).reshape(-1, 3)
        tm = np.ascontiguousarray(sd.tri_mesh, np.uint32)
        mats = _f32(sd.materials, (-1, 8))
        sph = _f32(sd.spheres, (-1, 5))
        _check(
            lib().cgrt_scene_create(
                _ptr(pn), len(pn), _ptr(tri), _ptr(tm), len(tri), _ptr(mats), len(mats), _ptr(sph) if len(sph) else None,
                len(sph), device, C.byref(self._h),
            )
        )  # fmt: skip
        self.device = device

    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            lib().cgrt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self) -> None:
        if lib is not None and C is not None:  # module globals are already gone when the interpreter is shutting down
            self.close()

    def hint_counts(self):
        """Lengths of the three rotating hard lists of the primary frames' hints (diagnostics)."""
        out = (C.c_uint32 * 3)()
        _check(lib().cgrt_debug_hint_counts(self._h, out))
        return [int(v) for v in out]

    def last_render_path(self) -> int:
        """0 = the last render() sized every list exactly, 1 = drawn as the previous frame predicted, 2 = predicted, too small, drawn again."""
        return int(lib().cgrt_debug_render_path(self._h))

    # ---- certified walk (same results as the exact walk) ----
    def set_walk(self, certified: bool) -> None:
        _check(lib().cgrt_scene_set_walk(self._h, 1 if certified else 0))

    def walk(self) -> int:
        """1 = certified walk (fast tree + certificate, exact walk as fallback), 0 = exact walk only."""
        return int(lib().cgrt_scene_walk(self._h))

    def build_info(self) -> dict:
        """What the host builder decided: fast tree built?, leaves with a degenerate float plane, finite geometry?, leaf count."""
        out = np.zeros(4, np.uint32)
        _check(lib().cgrt_scene_build_info(self._h, _ptr(out)))
        return dict(fast_tree=bool(out[0]), wild_leaves=int(out[1]), geometry_finite=bool(out[2]), leaves=int(out[3]))

    # ---- introspection ----
    def num_levels(self) -> int:
        return int(lib().cgrt_num_levels(self._h))

    def num_subnodes(self) -> int:
        return int(lib().cgrt_num_subnodes(self._h))

    def build_seconds(self) -> float:
        return float(lib().cgrt_build_seconds(self._h))

    def device_bytes(self) -> int:
        return int(lib().cgrt_device_bytes(self._h))

    def nodes(self) -> Tuple[np.ndarray, np.ndarray]:
        n = int(lib().cgrt_num_nodes(self._h))
        meta = np.zeros((n, 5), np.int32)
        boxes = np.zeros((n, 6), np.float32)
        if n:
            _check(lib().cgrt_get_nodes(self._h, _ptr(meta), _ptr(boxes)))
        return meta, boxes

    def leaf_prims(self, node: int) -> np.ndarray:
        n = int(lib().cgrt_leaf_prims(self._h, node, None, 0))
        out = np.zeros(n, np.uint32)
        if n:
            lib().cgrt_leaf_prims(self._h, node, _ptr(out), n)
        return out

    # ---- hot path ----
    def intersect(self, rays: np.ndarray, want_normals: bool = True):
        """Batched BoundingVolumeHierarchy::intersect. Returns (hits[HIT_DTYPE], normals or None)."""
        rays = _as_ray_array(rays)
        hits = np.zeros(len(rays), HIT_DTYPE)
        normals = np.zeros((len(rays), 3), np.float32) if want_normals else None
        _check(lib().cgrt_intersect_batch(self._h, _ptr(rays), len(rays), _ptr(hits), _ptr(normals)))
        return hits, normals

    def intersect_brute(self, rays: np.ndarray, mesh: int = -1, want_normals: bool = True):
        """cgrt_intersect_brute_batch: every triangle, no tree (ray_tracing.cpp:202-213). mesh < 0: all meshes + spheres."""
        rays = _as_ray_array(rays)
        hits = np.zeros(len(rays), HIT_DTYPE)
        normals = np.zeros((len(rays), 3), np.float32) if want_normals else None
        _check(lib().cgrt_intersect_brute_batch(self._h, _ptr(rays), len(rays), mesh, _ptr(hits), _ptr(normals)))
        return hits, normals

    def trace_primary(self, cam, W: int, H: int, rect=None, rank: int = 0, nranks: int = 1, want_normals: bool = False):
        x0, y0, x1, y1 = rect if rect is not None else (0, 0, W, H)
        hits = np.zeros(W * H, HIT_DTYPE)
        hits["prim_id"] = NO_PRIM
        hits["material_id"] = -1
        hits["t"] = np.nan  # pixels this call does not own stay marked
        normals = np.zeros((W * H, 3), np.float32) if want_normals else None
        c = cam if isinstance(cam, Camera) else Camera.from_array(cam)
        _check(lib().cgrt_trace_primary(self._h, C.byref(c), W, H, x0, y0, x1, y1, rank, nranks, _ptr(hits), _ptr(normals)))
        return hits, normals

    def trace_primary_device(self, cam, W, H, d_hits_ptr: int, rect=None, rank=0, nranks=1, d_normals_ptr: int = 0, stream: int = 0):
        x0, y0, x1, y1 = rect if rect is not None else (0, 0, W, H)
        c = cam if isinstance(cam, Camera) else Camera.from_array(cam)
        _check(
            lib().cgrt_trace_primary_device(
                self._h, C.byref(c), W, H, x0, y0, x1, y1, rank, nranks, C.c_void_p(d_hits_ptr),
                C.c_void_p(d_normals_ptr) if d_normals_ptr else None, C.c_void_p(stream) if stream else None,
            )
        )  # fmt: skip

    def intersect_device(self, d_rays_ptr: int, n: int, d_hits_ptr: int, d_normals_ptr: int = 0, stream: int = 0):
        _check(
            lib().cgrt_intersect_batch_device(
                self._h, C.c_void_p(d_rays_ptr), n, C.c_void_p(d_hits_ptr), C.c_void_p(d_normals_ptr) if d_normals_ptr else None,
                C.c_void_p(stream) if stream else None,
            )
        )  # fmt: skip

    def check_layout(self) -> None:
        """cgrt_debug_check_layout: raises when a reference of the record arrays is inconsistent."""
        _check(lib().cgrt_debug_check_layout(self._h))

    def layout_hash(self) -> int:
        """cgrt_debug_layout_hash: FNV-1a over every array the device reads."""
        h = C.c_uint64()
        _check(lib().cgrt_debug_layout_hash(self._h, C.byref(h)))
        return int(h.value)

    def render(self, cam, W: int, H: int, lights=None, max_level: int = 2):
        """cgrt_render: the whole shading/recursion driver on the device. Returns (rgb[W*H,3], stats dict)."""
        lights = _f32(self.sd.point_lights if lights is None else lights, (-1, 6))
        rgb = np.zeros((W * H, 3), np.float32)
        st = RenderStats()
        c = cam if isinstance(cam, Camera) else Camera.from_array(cam)
        _check(lib().cgrt_render(self._h, C.byref(c), W, H, _ptr(lights), len(lights), max_level, _ptr(rgb), C.byref(st)))
        return rgb, {k: getattr(st, k) for k, _ in st._fields_}

    def render_mapped(self, cam, W: int, H: int, lights=None, max_level: int = 2):
        """cgrt_render_mapped: the frame stays in the scene's pinned staging memory; returns (a COPY of it as rgb[W*H,3], stats)
        -- the copy is for the caller's convenience here, the C caller reads the pinned frame in place."""
        lights = _f32(self.sd.point_lights if lights is None else lights, (-1, 6))
        st = RenderStats()
        ptr = C.c_void_p()
        c = cam if isinstance(cam, Camera) else Camera.from_array(cam)
        _check(lib().cgrt_render_mapped(self._h, C.byref(c), W, H, _ptr(lights), len(lights), None, max_level, C.byref(ptr), C.byref(st)))
        rgb = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(W * H, 3)).copy()
        return rgb, {k: getattr(st, k) for k, _ in st._fields_}

    def render_counted(self, cam, W: int, H: int, lights=None, max_level: int = 2):
        """cgrt_render_counted: (rgb, stats, {"primary" / "shadow" / "mirror": counters dict}) of an instrumented frame."""
        lights = _f32(self.sd.point_lights if lights is None else lights, (-1, 6))
        rgb = np.zeros((W * H, 3), np.float32)
        st = RenderStats()
        work = (Counters * 3)()
        c = cam if isinstance(cam, Camera) else Camera.from_array(cam)
        _check(lib().cgrt_render_counted(self._h, C.byref(c), W, H, _ptr(lights), len(lights), max_level, _ptr(rgb), C.byref(st), work))
        return rgb, {k: getattr(st, k) for k, _ in st._fields_}, dict(zip(("primary", "shadow", "mirror"), (w.as_dict() for w in work)))

    def render_soft(self, cam, W: int, H: int, spherical, units, samples: int = 200, seed: int = 0, lights=None, max_level: int = 2,
                    closest_hit: bool = False):
        """cgrt_render_soft: + spherical lights (n x 7 {position, radius, color}) sampled with `samples` shadow rays per hit
        (main.cpp:168-218); `units` (n x 3) are the randomUnitVector() draws, see unit_vector_table()."""
        lights = _f32(self.sd.point_lights if lights is None else lights, (-1, 6))
        spherical = _f32(spherical, (-1, 7))
        units = _f32(units, (-1, 3))
        rgb = np.zeros((W * H, 3), np.float32)
        st = RenderStats()
        q = SoftShadows(spherical.ctypes.data, units.ctypes.data, len(spherical), samples, len(units), seed, int(closest_hit))
        c = cam if isinstance(cam, Camera) else Camera.from_array(cam)
        _check(lib().cgrt_render_soft(self._h, C.byref(c), W, H, _ptr(lights), len(lights), C.byref(q), max_level, _ptr(rgb), C.byref(st)))
        return rgb, {k: getattr(st, k) for k, _ in st._fields_}

    def render_rank(self, cam, W: int, H: int, rank: int, nranks: int, rgb: Optional[np.ndarray] = None, lights=None, max_level: int = 2,
                    spherical=None, units=None, samples: int = 200, seed: int = 0):
        """cgrt_render_rank: this rank's super-tiles of the frame (shading included); other pixels of `rgb` are kept."""
        lights = _f32(self.sd.point_lights if lights is None else lights, (-1, 6))
        rgb = np.zeros((W * H, 3), np.float32) if rgb is None else rgb
        assert rgb.dtype == np.float32 and rgb.size == W * H * 3 and rgb.flags.c_contiguous
        st = RenderStats()
        q = None
        if spherical is not None:
            spherical, units = _f32(spherical, (-1, 7)), _f32(units, (-1, 3))
            q = C.byref(SoftShadows(spherical.ctypes.data, units.ctypes.data, len(spherical), samples, len(units), seed, 0))
        c = cam if isinstance(cam, Camera) else Camera.from_array(cam)
        _check(lib().cgrt_render_rank(self._h, C.byref(c), W, H, _ptr(lights), len(lights), q, max_level, rank, nranks, _ptr(rgb), C.byref(st)))
        return rgb, {k: getattr(st, k) for k, _ in st._fields_}

    def generate_rays(self, cam, W: int, H: int, rect=None) -> np.ndarray:
        x0, y0, x1, y1 = rect if rect is not None else (0, 0, W, H)
        rays = np.zeros((x1 - x0) * (y1 - y0), RAY_DTYPE)
        c = cam if isinstance(cam, Camera) else Camera.from_array(cam)
        _check(lib().cgrt_generate_rays(self._h, C.byref(c), W, H, x0, y0, x1, y1, _ptr(rays)))
        return rays

    def count_primary(self, cam, W, H, rect=None, rank=0, nranks=1) -> dict:
        x0, y0, x1, y1 = rect if rect is not None else (0, 0, W, H)
        c = cam if isinstance(cam, Camera) else Camera.from_array(cam)
        out = Counters()
        _check(lib().cgrt_count_primary(self._h, C.byref(c), W, H, x0, y0, x1, y1, rank, nranks, C.byref(out)))
        return out.as_dict()

    def debug_wave_times(self, cam, W, H) -> np.ndarray:
        """(ntiles, 16) u64 per wave, see cgrt_debug_wave_times in include/cgrt.h (diagnostic launch)."""
        nst = ((W + 63) // 64) * ((H + 63) // 64)
        nt = 4 * 16 * 8 * ((nst + 7) // 8)
        out = np.zeros((nt, 16), np.uint64)
        c = cam if isinstance(cam, Camera) else Camera.from_array(cam)
        _check(lib().cgrt_debug_wave_times(self._h, C.byref(c), W, H, _ptr(out), nt))
        return out

    def count_batch(self, rays: np.ndarray) -> dict:
        rays = _as_ray_array(rays)
        out = Counters()
        _check(lib().cgrt_count_batch(self._h, _ptr(rays), len(rays), C.byref(out)))
        return out.as_dict()


# ---- one caller, N devices, one framebuffer ----
def trace_primary_multi(scenes, cam, W: int, H: int, want_normals: bool = False):
    """cgrt_trace_primary_multi over replicas `scenes` (one Scene per device / rank). Returns (hits, normals or None, stats dict)."""
    arr = (C.c_void_p * len(scenes))(*[s._h for s in scenes])
    hits = np.zeros(W * H, HIT_DTYPE)
    normals = np.zeros((W * H, 3), np.float32) if want_normals else None
    st = MultiStats()
    c = cam if isinstance(cam, Camera) else Camera.from_array(cam)
    _check(lib().cgrt_trace_primary_multi(arr, len(scenes), C.byref(c), W, H, _ptr(hits), _ptr(normals), C.byref(st)))
    return hits, normals, dict(replicas=st.replicas, kernel_ms_max=st.kernel_ms_max, download_ms_max=st.download_ms_max, wall_ms=st.wall_ms,
                               rays=[int(st.rays[i]) for i in range(len(scenes))])


def render_multi(scenes, cam, W: int, H: int, lights=None, max_level: int = 2, spherical=None, units=None, samples: int = 200, seed: int = 0):
    """cgrt_render_multi over replicas `scenes`. Returns (rgb[W*H,3], stats dict)."""
    arr = (C.c_void_p * len(scenes))(*[s._h for s in scenes])
    lights = _f32(scenes[0].sd.point_lights if lights is None else lights, (-1, 6))
    rgb = np.zeros((W * H, 3), np.float32)
    st = RenderStats()
    q = None
    if spherical is not None:
        spherical, units = _f32(spherical, (-1, 7)), _f32(units, (-1, 3))
        q = C.byref(SoftShadows(spherical.ctypes.data, units.ctypes.data, len(spherical), samples, len(units), seed, 0))
    c = cam if isinstance(cam, Camera) else Camera.from_array(cam)
    _check(lib().cgrt_render_multi(arr, len(scenes), C.byref(c), W, H, _ptr(lights), len(lights), q, max_level, _ptr(rgb), C.byref(st)))
    return rgb, {k: getattr(st, k) for k, _ in st._fields_}


# ---- element-wise primitives (src/ray_tracing.h:10-20) ----
def ray_triangle(tri18, rays, device=0):
    tri18 = _f32(tri18, (-1, 18))
    rays = np.ascontiguousarray(rays, RAY_DTYPE)
    n = len(rays)
    t, hit, nrm = np.zeros(n, np.float32), np.zeros(n, np.uint8), np.zeros((n, 3), np.float32)
    _check(lib().cgrt_ray_triangle_batch(device, _ptr(tri18), _ptr(rays), n, _ptr(t), _ptr(hit), _ptr(nrm)))
    return t, hit, nrm


def ray_plane(plane4, rays, device=0):
    plane4 = _f32(plane4, (-1, 4))
    rays = np.ascontiguousarray(rays, RAY_DTYPE)
    n = len(rays)
    t, hit = np.zeros(n, np.float32), np.zeros(n, np.uint8)
    _check(lib().cgrt_ray_plane_batch(device, _ptr(plane4), _ptr(rays), n, _ptr(t), _ptr(hit)))
    return t, hit


def ray_box(box6, rays, device=0):
    box6 = _f32(box6, (-1, 6))
    rays = np.ascontiguousarray(rays, RAY_DTYPE)
    n = len(rays)
    t, hit, inside = np.zeros(n, np.float32), np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    _check(lib().cgrt_ray_box_batch(device, _ptr(box6), _ptr(rays), n, _ptr(t), _ptr(hit), _ptr(inside)))
    return t, hit, inside


def ray_sphere(sph4, rays, device=0):
    sph4 = _f32(sph4, (-1, 4))
    rays = np.ascontiguousarray(rays, RAY_DTYPE)
    n = len(rays)
    t, hit, nrm = np.zeros(n, np.float32), np.zeros(n, np.uint8), np.zeros((n, 3), np.float32)
    _check(lib().cgrt_ray_sphere_batch(device, _ptr(sph4), _ptr(rays), n, _ptr(t), _ptr(hit), _ptr(nrm)))
    return t, hit, nrm


def fastdiv_check(a, d, device=0):
    """(mismatch count, first_bad[a, d, got, expected]) of the kernels' exact fast division vs IEEE a / d."""
    a = np.ascontiguousarray(a, np.float32).reshape(-1)
    d = np.ascontiguousarray(d, np.float32).reshape(-1)
    mm = np.zeros(1, np.uint64)
    bad = np.zeros(4, np.float32)
    _check(lib().cgrt_debug_fastdiv_check(device, _ptr(a), _ptr(d), len(a), _ptr(mm), _ptr(bad)))
    return int(mm[0]), bad


def triangle_plane(tri9, device=0):
    tri9 = _f32(tri9, (-1, 9))
    out = np.zeros((len(tri9), 4), np.float32)
    _check(lib().cgrt_triangle_plane_batch(device, _ptr(tri9), len(tri9), _ptr(out)))
    return out


def point_in_triangle(in15, device=0):
    in15 = _f32(in15, (-1, 15))
    out = np.zeros(len(in15), np.uint8)
    _check(lib().cgrt_point_in_triangle_batch(device, _ptr(in15), len(in15), _ptr(out)))
    return out


# ---- C++ host mirror (cg-raytracer_amd/host): render driver, OBJ loader, BMP writer ----
_host = None


def host_lib() -> C.CDLL:
    global _host
    if _host is None:
        lib()  # libcgrt.so first (dependency)
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError(f"{HOST_LIB_PATH} is missing: run __graft_entry__.build()")
        H = C.CDLL(HOST_LIB_PATH)
        vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
        H.cgrt_host_last_error.restype = C.c_char_p
        H.cgrt_host_render.argtypes = [vp, u32, vp, vp, u32, vp, u32, vp, u32, vp, i32, i32, i32, vp, vp]
        H.cgrt_host_time_screen_render.argtypes = [vp, u32, vp, vp, u32, vp, u32, vp, u32, vp, i32, i32, i32, i32, vp]
        H.cgrt_host_render_per_ray.argtypes = [vp, u32, vp, vp, u32, vp, u32, vp, u32, vp, i32, i32, i32, i32, vp, vp]
        H.cgrt_host_render_soft.argtypes = [vp, u32, vp, vp, u32, vp, u32, vp, u32, vp, u32, vp, u32, u32, u32, vp, i32, i32, i32, vp, vp, i32]
        H.cgrt_host_load_obj.argtypes = [C.c_char_p, i32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), vp, vp, vp, vp]
        H.cgrt_host_write_bmp.argtypes = [C.c_char_p, vp, i32, i32]
        H.cgrt_host_selftest.argtypes = [vp, u32, vp, vp, u32, vp, u32, vp, u32, C.POINTER(i32)]
        H.cgrt_host_threads_test.argtypes = [vp, u32, vp, vp, u32, vp, u32, vp, u32, i32, vp]
        H.cgrt_host_render_bmp.argtypes = [vp, u32, vp, vp, u32, vp, u32, vp, u32, vp, i32, i32, i32, i32, C.c_char_p, vp]
        _host = H
    return _host


def host_render(sd: SceneData, cam, W: int, H: int, max_level: int = 2):
    """renderRayTracing of the C++ host mirror (wavefront over the GPU path). Returns (rgb[H*W,3], stats dict)."""
    Hl = host_lib()
    pn, tri = _f32(sd.pos_nrm, (-1, 6)), np.ascontiguousarray(sd.tri, np.uint32).reshape(-1, 3)
    tm, mats = np.ascontiguousarray(sd.tri_mesh, np.uint32), _f32(sd.materials, (-1, 8))
    lights, camv = _f32(sd.point_lights, (-1, 6)), _f32(cam, (9,))
    rgb = np.zeros((W * H, 3), np.float32)
    st = np.zeros(5, np.float64)
    rc = Hl.cgrt_host_render(_ptr(pn), len(pn), _ptr(tri), _ptr(tm), len(tri), _ptr(mats), len(mats), _ptr(lights), len(lights),
                             _ptr(camv), W, H, max_level, _ptr(rgb), _ptr(st))
    if rc:
        raise RuntimeError("cgrt_host_render: " + Hl.cgrt_host_last_error().decode())
    return rgb, dict(primary=int(st[0]), shadow=int(st[1]), reflection=int(st[2]), seconds_device=float(st[3]), seconds_total=float(st[4]))


def host_time_screen_render(sd: SceneData, cam, W: int, H: int, max_level: int = 2, reps: int = 5) -> dict:
    """Whole-call milliseconds of the mirror's Screen-filling drivers (BVH built once): renderRayTracingOnDevice, renderRayTracing."""
    Hl = host_lib()
    pn, tri = _f32(sd.pos_nrm, (-1, 6)), np.ascontiguousarray(sd.tri, np.uint32).reshape(-1, 3)
    tm, mats = np.ascontiguousarray(sd.tri_mesh, np.uint32), _f32(sd.materials, (-1, 8))
    lights, camv = _f32(sd.point_lights, (-1, 6)), _f32(cam, (9,))
    ms = np.zeros(3, np.float64)
    rc = Hl.cgrt_host_time_screen_render(_ptr(pn), len(pn), _ptr(tri), _ptr(tm), len(tri), _ptr(mats), len(mats), _ptr(lights), len(lights),
                                         _ptr(camv), W, H, max_level, reps, _ptr(ms))
    if rc:
        raise RuntimeError("cgrt_host_time_screen_render: " + Hl.cgrt_host_last_error().decode())
    return dict(on_device_ms=float(ms[0]), host_wavefront_ms=float(ms[1]), on_device_device_share_ms=float(ms[2]))


def host_render_per_ray(sd: SceneData, cam, W: int, H: int, max_level: int = 2, threads: int = 0):
    """The reference's driver taken literally through the C++ mirror (renderToBufferPerRay): omp parallel for over rows, per-pixel
    recursion, one BoundingVolumeHierarchy::intersect call per ray from `threads` threads. Returns (rgb[H*W,3], stats dict)."""
    Hl = host_lib()
    pn, tri = _f32(sd.pos_nrm, (-1, 6)), np.ascontiguousarray(sd.tri, np.uint32).reshape(-1, 3)
    tm, mats = np.ascontiguousarray(sd.tri_mesh, np.uint32), _f32(sd.materials, (-1, 8))
    lights, camv = _f32(sd.point_lights, (-1, 6)), _f32(cam, (9,))
    rgb = np.zeros((W * H, 3), np.float32)
    st = np.zeros(5, np.float64)
    rc = Hl.cgrt_host_render_per_ray(_ptr(pn), len(pn), _ptr(tri), _ptr(tm), len(tri), _ptr(mats), len(mats), _ptr(lights), len(lights),
                                     _ptr(camv), W, H, max_level, threads, _ptr(rgb), _ptr(st))
    if rc:
        raise RuntimeError("cgrt_host_render_per_ray: " + Hl.cgrt_host_last_error().decode())
    return rgb, dict(primary=int(st[0]), shadow=int(st[1]), reflection=int(st[2]), seconds_device=float(st[3]), seconds_total=float(st[4]))


def host_render_soft(sd: SceneData, cam, W: int, H: int, spherical, units=None, samples: int = 200, seed: int = 0, lights=None,
                     max_level: int = 2, on_device: bool = False):
    """renderRayTracing of the C++ host mirror for a scene with spherical lights (soft shadows, main.cpp:168-218).
    units=None uses the mirror's own gaussian table (SoftShadowSampler::gaussian); on_device=True runs the C++ mirror's
    renderToBufferOnDevice (the whole driver on the GPU) instead of its host-driven wavefront."""
    Hl = host_lib()
    pn, tri = _f32(sd.pos_nrm, (-1, 6)), np.ascontiguousarray(sd.tri, np.uint32).reshape(-1, 3)
    tm, mats = np.ascontiguousarray(sd.tri_mesh, np.uint32), _f32(sd.materials, (-1, 8))
    lights, camv = _f32(sd.point_lights if lights is None else lights, (-1, 6)), _f32(cam, (9,))
    spherical = _f32(spherical, (-1, 7))
    units = np.zeros((0, 3), np.float32) if units is None else _f32(units, (-1, 3))
    rgb = np.zeros((W * H, 3), np.float32)
    st = np.zeros(6, np.float64)
    rc = Hl.cgrt_host_render_soft(_ptr(pn), len(pn), _ptr(tri), _ptr(tm), len(tri), _ptr(mats), len(mats), _ptr(lights), len(lights),
                                  _ptr(spherical), len(spherical), _ptr(units), len(units), samples, seed, _ptr(camv), W, H, max_level,
                                  _ptr(rgb), _ptr(st), int(on_device))
    if rc:
        raise RuntimeError("cgrt_host_render_soft: " + Hl.cgrt_host_last_error().decode())
    return rgb, dict(primary=int(st[0]), shadow=int(st[1]), reflection=int(st[2]), soft_shadow=int(st[3]), seconds_device=float(st[4]),
                     seconds_total=float(st[5]))


def host_load_obj(path: str, normalize: bool = False) -> SceneData:
    """loadMesh of the C++ host mirror, as flat arrays."""
    Hl = host_lib()
    nv, nt, nm = C.c_uint32(), C.c_uint32(), C.c_uint32()
    if Hl.cgrt_host_load_obj(path.encode(), int(normalize), C.byref(nv), C.byref(nt), C.byref(nm), None, None, None, None):
        raise RuntimeError("cgrt_host_load_obj: " + Hl.cgrt_host_last_error().decode())
    pn = np.zeros((nv.value, 6), np.float32)
    tri = np.zeros((nt.value, 3), np.uint32)
    tm = np.zeros(nt.value, np.uint32)
    mats = np.zeros((nm.value, 8), np.float32)
    Hl.cgrt_host_load_obj(path.encode(), int(normalize), C.byref(nv), C.byref(nt), C.byref(nm), _ptr(pn), _ptr(tri), _ptr(tm), _ptr(mats))
    return SceneData(pos_nrm=pn, tri=tri, tri_mesh=tm, materials=mats)


def host_selftest(sd: SceneData, rays7) -> Tuple[int, int]:
    """Runs the C++ mirror's per-ray API (BoundingVolumeHierarchy::intersect, free functions, copy-assign) against its
    batched API on the given rays. Returns (disagreements, numLevels)."""
    Hl = host_lib()
    pn, tri = _f32(sd.pos_nrm, (-1, 6)), np.ascontiguousarray(sd.tri, np.uint32).reshape(-1, 3)
    tm, mats = np.ascontiguousarray(sd.tri_mesh, np.uint32), _f32(sd.materials, (-1, 8))
    r = _f32(rays7, (-1, 7))
    lv = C.c_int(0)
    bad = Hl.cgrt_host_selftest(_ptr(pn), len(pn), _ptr(tri), _ptr(tm), len(tri), _ptr(mats), len(mats), _ptr(r), len(r), C.byref(lv))
    if bad < 0:
        raise RuntimeError("cgrt_host_selftest: " + Hl.cgrt_host_last_error().decode())
    return int(bad), int(lv.value)


def host_threads_test(sd: SceneData, rays7, nthreads: int = 8):
    """nthreads std::threads call BoundingVolumeHierarchy::intersect per ray on ONE BVH object (as main.cpp:653-656 does);
    returns (disagreements with intersectBatch, timing dict)."""
    Hl = host_lib()
    pn, tri = _f32(sd.pos_nrm, (-1, 6)), np.ascontiguousarray(sd.tri, np.uint32).reshape(-1, 3)
    tm, mats = np.ascontiguousarray(sd.tri_mesh, np.uint32), _f32(sd.materials, (-1, 8))
    r = _f32(rays7, (-1, 7))
    tim = np.zeros(8, np.float64)
    bad = Hl.cgrt_host_threads_test(_ptr(pn), len(pn), _ptr(tri), _ptr(tm), len(tri), _ptr(mats), len(mats), _ptr(r), len(r), nthreads, _ptr(tim))
    if bad < 0:
        raise RuntimeError("cgrt_host_threads_test: " + Hl.cgrt_host_last_error().decode())
    return int(bad), dict(us_per_call_one_thread=float(tim[0]), us_per_call_per_thread=float(tim[1]), calls_per_second=float(tim[2]),
                          combined_generations=int(tim[3]), combined_rays=int(tim[4]), largest_generation=int(tim[5]),
                          leader_gpu_us_per_generation=float(tim[6]) / 1e3 / max(1.0, float(tim[3])),
                          leader_launch_us_per_generation=float(tim[7]) / 1e3 / max(1.0, float(tim[3])))


def host_render_bmp(sd: SceneData, cam, W: int, H: int, path: str, max_level: int = 2, nreplicas: int = 1):
    """Scene -> device render (nreplicas BVH replicas sharing the frame) -> Screen -> BMP file; returns the float frame (W*H, 3)."""
    Hl = host_lib()
    pn, tri = _f32(sd.pos_nrm, (-1, 6)), np.ascontiguousarray(sd.tri, np.uint32).reshape(-1, 3)
    tm, mats = np.ascontiguousarray(sd.tri_mesh, np.uint32), _f32(sd.materials, (-1, 8))
    lights, camv = _f32(sd.point_lights, (-1, 6)), _f32(cam, (9,))
    rgb = np.zeros((W * H, 3), np.float32)
    rc = Hl.cgrt_host_render_bmp(_ptr(pn), len(pn), _ptr(tri), _ptr(tm), len(tri), _ptr(mats), len(mats), _ptr(lights), len(lights), _ptr(camv),
                                 W, H, max_level, nreplicas, path.encode(), _ptr(rgb))
    if rc:
        raise RuntimeError("cgrt_host_render_bmp: " + Hl.cgrt_host_last_error().decode())
    return rgb


def host_write_bmp(path: str, rgb, W: int, H: int) -> None:
    rgb = _f32(rgb, (W * H, 3))
    if host_lib().cgrt_host_write_bmp(path.encode(), _ptr(rgb), W, H):
        raise RuntimeError("cgrt_host_write_bmp: " + host_lib().cgrt_host_last_error().decode())
