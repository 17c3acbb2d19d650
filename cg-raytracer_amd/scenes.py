"""Scene data for tests and bench: OBJ/MTL loading with assimp-like semantics, the reference's scene
presets, and the procedural stand-in for the absent dragon.obj.

Nothing here is on the hot path; it produces the flat arrays the C-ABI takes
(``cgrt_scene_create``: pos_nrm V x 6, tri T x 3, tri_mesh T, materials M x 8, spheres S x 5).

Reference behaviour restated (paths relative to /root/reference):
  * src/mesh.cpp:58-141  loadMesh: assimp 5.0.1 with aiProcess_GenNormals | aiProcess_Triangulate and
    no vertex joining -> one vertex per face corner, one Mesh per (object/group, material) run, the node
    tree walked with a LIFO stack (so sibling objects come out in reverse file order).
  * src/mesh.cpp:143-166 centerAndScaleToUnitMesh: mean of all vertex positions (sequential float32
    accumulation), then divide by the largest distance to it.
  * src/scene.cpp:4-69   the presets (which file, normalise or not, lights, spheres).
assimp itself is not in the image, so loader parity is unpinned; the hot path never
sees files and does not depend on it.
"""
from __future__ import annotations

import dataclasses
import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np

F32 = np.float32


@dataclasses.dataclass
class SceneData:
    """Flat scene arrays in the layout of include/cgrt.h."""

    pos_nrm: np.ndarray  # (V, 6) float32
    tri: np.ndarray  # (T, 3) uint32, global vertex indices
    tri_mesh: np.ndarray  # (T,) uint32, non-decreasing
    materials: np.ndarray  # (M, 8) float32: kd ks shininess transparency
    spheres: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros((0, 5), F32))
    point_lights: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros((0, 6), F32))
    name: str = ""

    @property
    def ntris(self) -> int:
        return int(self.tri.shape[0])

    @property
    def nmesh(self) -> int:
        return int(self.materials.shape[0])

    def save(self, path: str) -> None:
        np.savez_compressed(
            path,
            pos_nrm=self.pos_nrm,
            tri=self.tri,
            tri_mesh=self.tri_mesh,
            materials=self.materials,
            spheres=self.spheres,
            point_lights=self.point_lights,
        )

    @staticmethod
    def load(path: str) -> "SceneData":
        z = np.load(path, allow_pickle=False)
        return SceneData(
            pos_nrm=z["pos_nrm"].astype(F32),
            tri=z["tri"].astype(np.uint32),
            tri_mesh=z["tri_mesh"].astype(np.uint32),
            materials=z["materials"].astype(F32),
            spheres=z["spheres"].astype(F32),
            point_lights=z["point_lights"].astype(F32),
            name=os.path.splitext(os.path.basename(path))[0],
        )


# --------------------------------------------------------------------------------------------------
# OBJ / MTL
# --------------------------------------------------------------------------------------------------
def _parse_mtl(path: str) -> Dict[str, np.ndarray]:
    """name -> 8 floats.  Defaults follow assimp's ObjFile::Material: Kd 0.6, Ks 0, Ns 0, d 1."""
    mats: Dict[str, np.ndarray] = {}
    cur: Optional[np.ndarray] = None
    if not os.path.exists(path):
        return mats
    with open(path, "r", errors="replace") as f:
        for line in f:
            p = line.split()
            if not p or p[0].startswith("#"):
                continue
            if p[0] == "newmtl":
                cur = np.array([0.6, 0.6, 0.6, 0, 0, 0, 0.0, 1.0], F32)
                mats[" ".join(p[1:])] = cur
            elif cur is not None:
                if p[0] == "Kd":
                    cur[0:3] = [float(x) for x in p[1:4]]
                elif p[0] == "Ks":
                    cur[3:6] = [float(x) for x in p[1:4]]
                elif p[0] == "Ns":
                    cur[6] = float(p[1])
                elif p[0] == "d":
                    cur[7] = float(p[1])
    return mats


def _face_normal(a: np.ndarray, b: np.ndarray, c: np.ndarray) -> np.ndarray:
    n = np.cross((b - a).astype(F32), (c - a).astype(F32)).astype(F32)
    l = F32(math.sqrt(float(np.dot(n, n))))
    return (n / l).astype(F32) if l > 0 else np.zeros(3, F32)


def load_obj(path: str, normalize: bool = False) -> SceneData:
    """loadMesh(file, normalize) (src/mesh.cpp:58-141) without assimp."""
    verts: List[List[float]] = []
    norms: List[List[float]] = []
    mtl: Dict[str, np.ndarray] = {}
    # objects in file order; each object = list of (material name, [faces]) runs; face = list of (v, vn) indices
    objects: List[List[Tuple[str, list]]] = [[]]
    cur_mat = "__default__"

    def cur_run() -> list:
        obj = objects[-1]
        if not obj or obj[-1][0] != cur_mat:
            obj.append((cur_mat, []))
        return obj[-1][1]

    with open(path, "r", errors="replace") as f:
        for line in f:
            p = line.split()
            if not p or p[0].startswith("#"):
                continue
            k = p[0]
            if k == "v":
                verts.append([float(x) for x in p[1:4]])
            elif k == "vn":
                norms.append([float(x) for x in p[1:4]])
            elif k == "mtllib":
                mtl.update(_parse_mtl(os.path.join(os.path.dirname(path), " ".join(p[1:]))))
            elif k in ("o", "g"):
                objects.append([])  # assimp maps groups onto objects (ObjFileParser::getGroupName)
            elif k == "usemtl":
                cur_mat = " ".join(p[1:])
            elif k == "f":
                face = []
                for tok in p[1:]:
                    q = tok.split("/")
                    vi = int(q[0])
                    vi = vi - 1 if vi > 0 else len(verts) + vi
                    ni = -1
                    if len(q) >= 3 and q[2]:
                        ni = int(q[2])
                        ni = ni - 1 if ni > 0 else len(norms) + ni
                    face.append((vi, ni))
                if len(face) >= 3:
                    cur_run().append(face)

    V = np.asarray(verts, F32).reshape(-1, 3)
    N = np.asarray(norms, F32).reshape(-1, 3)
    mesh_list = []  # (pos_nrm rows, local tris, material)
    # mesh.cpp:77-131: children are pushed on a stack and popped last-first => reverse object order;
    # the meshes of one node stay in order.
    for obj in reversed([o for o in objects if any(len(r[1]) for r in o)]):
        for mat_name, faces in obj:
            if not faces:
                continue
            rows: List[np.ndarray] = []
            tris: List[Tuple[int, int, int]] = []
            for face in faces:
                base = len(rows)
                P = [V[vi] for vi, _ in face]
                has_n = all(ni >= 0 for _, ni in face)
                fn = None if has_n else _face_normal(P[0], P[1], P[2])  # aiProcess_GenNormals: flat
                for (vi, ni), pos in zip(face, P):
                    rows.append(np.concatenate([pos, N[ni] if has_n else fn]).astype(F32))
                # aiProcess_Triangulate: fan from the first corner for convex polygons
                for k in range(1, len(face) - 1):
                    tris.append((base, base + k, base + k + 1))
            m = mtl.get(mat_name)
            if m is None:
                m = np.array([0.6, 0.6, 0.6, 0, 0, 0, 0.0, 1.0], F32)
            mesh_list.append((np.asarray(rows, F32), np.asarray(tris, np.uint32), m.copy()))

    pos_nrm = np.concatenate([m[0] for m in mesh_list]).astype(F32)
    off = 0
    tri_parts, tm_parts = [], []
    for i, (rows, tris, _) in enumerate(mesh_list):
        tri_parts.append(tris + np.uint32(off))
        tm_parts.append(np.full(len(tris), i, np.uint32))
        off += len(rows)
    sd = SceneData(
        pos_nrm=pos_nrm,
        tri=np.concatenate(tri_parts).astype(np.uint32),
        tri_mesh=np.concatenate(tm_parts).astype(np.uint32),
        materials=np.stack([m[2] for m in mesh_list]).astype(F32),
        name=os.path.splitext(os.path.basename(path))[0],
    )
    if normalize:
        center_and_scale_to_unit(sd)
    return sd


def center_and_scale_to_unit(sd: SceneData) -> None:
    """centerAndScaleToUnitMesh (src/mesh.cpp:143-166): std::accumulate in float32, then / maxD."""
    P = sd.pos_nrm[:, 0:3]
    acc = np.zeros(3, F32)
    # sequential float32 accumulation, as std::accumulate does; blocked cumsum keeps it exact-order
    for a in range(3):
        acc[a] = np.cumsum(P[:, a], dtype=F32)[-1] if len(P) else F32(0)
    center = (acc / F32(len(P))).astype(F32)
    dlt = (P - center).astype(F32)
    d2 = ((dlt[:, 0] * dlt[:, 0] + dlt[:, 1] * dlt[:, 1]).astype(F32) + dlt[:, 2] * dlt[:, 2]).astype(F32)
    max_d = F32(np.sqrt(d2.max()))
    sd.pos_nrm[:, 0:3] = (dlt / max_d).astype(F32)


# --------------------------------------------------------------------------------------------------
# Reference presets (src/scene.cpp:4-69) and camera default (src/main.cpp:730-731)
# --------------------------------------------------------------------------------------------------
PRESETS = {
    # name: (file, normalize, point lights [(pos, color)])
    "triangle": ("triangle.obj", False, [((-1, 1, -1), (1, 1, 1))]),
    "cube": ("cube.obj", False, [((-1, 1, -1), (1, 1, 1))]),
    "cornell": ("CornellBox-Mirror-Rotated.obj", True, [((0, 0.58, 0), (1, 1, 1))]),
    "monkey": ("monkey-rotated.obj", True, [((-1, 1, -1), (1, 1, 1)), ((1, -1, -1), (1, 1, 1))]),
    "dodge": ("dodgeColorTest.obj", False, [((-1, 1, -1), (1, 1, 1))]),
}


# SceneType::CornellBoxSphericalLight (src/scene.cpp:27-32): the Cornell box lit by ONE spherical light {position,
# radius, color} and no point light; sampled with 200 shadow rays per hit (main.cpp:176).
CORNELL_SPHERICAL_LIGHTS = np.asarray([[0.0, 0.45, 0.0, 0.1, 1.0, 1.0, 1.0]], F32)
SOFT_SHADOW_SAMPLES = 200


def load_preset(name: str, data_dir: str) -> SceneData:
    fn, norm, lights = PRESETS[name]
    sd = load_obj(os.path.join(data_dir, fn), normalize=norm)
    if name == "triangle":
        sd.materials[0, 0:3] = 1.0  # scene.cpp:11
    sd.point_lights = np.asarray([list(p) + list(c) for p, c in lights], F32).reshape(-1, 6)
    sd.name = name
    return sd


def spheres_preset() -> SceneData:
    """SceneType::Spheres (src/scene.cpp:51-56): three spheres, no meshes."""
    sp = np.asarray(
        [[3.0, -2.0, 10.2, 1.0, -1], [-2.0, 2.0, 4.0, 2.0, -1], [0.0, 0.0, 6.0, 0.75, -1]],
        F32,
    )
    return SceneData(
        pos_nrm=np.zeros((0, 6), F32),
        tri=np.zeros((0, 3), np.uint32),
        tri_mesh=np.zeros((0,), np.uint32),
        materials=np.zeros((0, 8), F32),
        spheres=sp,
        point_lights=np.asarray([[3, 0, 3, 15, 15, 15]], F32),
        name="spheres",
    )


def default_camera(width: int, height: int) -> np.ndarray:
    """look_at(3) euler_rad(3) distance fovy_rad aspect -- main.cpp:730-731, window.cpp:334-337.
    glm::radians(x) = x * 0.01745329251994329576923690768489f in float32."""
    k = F32(0.01745329251994329576923690768489)
    return np.asarray(
        [0, 0, 0, F32(20.0) * k, F32(20.0) * k, F32(0.0) * k, 3.0, F32(50.0) * k, F32(width) / F32(height)], F32
    )


# --------------------------------------------------------------------------------------------------
# Procedural stand-in for data/dragon.obj (absent: .MISSING_LARGE_BLOBS:1)
# --------------------------------------------------------------------------------------------------
def make_dragon(ntris_target: int = 800_000, seed: int = 20250117) -> SceneData:
    """Deterministic closed, bumpy, self-occluding surface with ~ntris_target triangles in ONE mesh,
    un-indexed the way assimp delivers OBJ data (3 vertices per triangle), smooth per-vertex normals,
    centred and unit-scaled like scene.cpp:42 does for dragon.obj.

    Shape: a (2,3) torus-knot tube whose radius is modulated by a few fixed sinusoids (ridges/scales),
    tessellated on a (nu x nv) grid.  Pure function of (ntris_target, seed); computed in float64 and
    rounded once to float32, so the same arrays come out on every machine with IEEE-correct sin/cos
    inputs -- and the committed test fixtures never depend on it being bit-stable across libms
    (full-size parity is GPU vs oracle on the SAME generated arrays).
    """
    rng = np.random.RandomState(seed)
    # grid: 2 * nu * nv triangles, tube is ~12x longer than around
    nv = max(8, int(round(math.sqrt(ntris_target / 2.0 / 12.0))))
    nu = max(16, int(round(ntris_target / 2.0 / nv)))
    u = np.linspace(0.0, 2.0 * np.pi, nu, endpoint=False)
    v = np.linspace(0.0, 2.0 * np.pi, nv, endpoint=False)
    p, q = 2.0, 3.0
    # knot centre line and a (non-twisting) frame by finite differences
    def centre(uu):
        r = 1.0 + 0.45 * np.cos(q * uu)
        return np.stack([r * np.cos(p * uu), r * np.sin(p * uu), 0.45 * np.sin(q * uu)], -1)

    C = centre(u)
    du = 1e-4
    T = centre(u + du) - centre(u - du)
    T /= np.linalg.norm(T, axis=-1, keepdims=True)
    up = np.array([0.0, 0.0, 1.0])
    Nn = np.cross(T, up)
    Nn /= np.linalg.norm(Nn, axis=-1, keepdims=True)
    Bn = np.cross(T, Nn)
    # radius field: base + ridges along the body + fine scales (fixed phases from the seed)
    ph = rng.uniform(0, 2 * np.pi, size=6)
    U, Vv = np.meshgrid(u, v, indexing="ij")
    rad = (
        0.16
        + 0.05 * np.sin(5.0 * U + ph[0])
        + 0.03 * np.sin(9.0 * U + 3.0 * Vv + ph[1])
        + 0.018 * np.sin(31.0 * U + ph[2]) * np.sin(7.0 * Vv + ph[3])
        + 0.008 * np.sin(97.0 * U + 13.0 * Vv + ph[4])
        + 0.004 * np.sin(211.0 * U + ph[5]) * np.cos(29.0 * Vv)
    )
    P = C[:, None, :] + rad[..., None] * (np.cos(Vv)[..., None] * Nn[:, None, :] + np.sin(Vv)[..., None] * Bn[:, None, :])
    # smooth normals from the grid (central differences, wraps in both directions)
    dPu = np.roll(P, -1, 0) - np.roll(P, 1, 0)
    dPv = np.roll(P, -1, 1) - np.roll(P, 1, 1)
    Nrm = np.cross(dPv, dPu)
    Nrm /= np.maximum(np.linalg.norm(Nrm, axis=-1, keepdims=True), 1e-30)
    # two triangles per cell, un-indexed
    i0 = np.arange(nu)[:, None]
    j0 = np.arange(nv)[None, :]
    i1 = (i0 + 1) % nu
    j1 = (j0 + 1) % nv

    def g(a, i, j):
        return a[np.broadcast_to(i, (nu, nv)), np.broadcast_to(j, (nu, nv))]

    quads = [(i0, j0), (i1, j0), (i1, j1), (i0, j1)]
    PA = [g(P, i, j) for i, j in quads]
    NA = [g(Nrm, i, j) for i, j in quads]
    corners = [(0, 1, 2), (0, 2, 3)]
    pos = np.stack([np.stack([PA[c] for c in tri], 2) for tri in corners], 2)  # nu nv 2 3 xyz
    nrm = np.stack([np.stack([NA[c] for c in tri], 2) for tri in corners], 2)
    pos = pos.reshape(-1, 3)
    nrm = nrm.reshape(-1, 3)
    ntri = pos.shape[0] // 3
    sd = SceneData(
        pos_nrm=np.concatenate([pos, nrm], 1).astype(F32),
        tri=np.arange(3 * ntri, dtype=np.uint32).reshape(ntri, 3),
        tri_mesh=np.zeros(ntri, np.uint32),
        materials=np.asarray([[0.8, 0.8, 0.8, 0.5, 0.5, 0.5, 225.0, 1.0]], F32),
        point_lights=np.asarray([[-1, 1, -1, 1, 1, 1]], F32),
        name=f"dragon{ntri}",
    )
    center_and_scale_to_unit(sd)
    return sd


def make_dragon_irregular(ntris_target: int = 800_000, seed: int = 20250117) -> SceneData:
    """The dragon stand-in with what a scanned mesh has and a tessellated grid has not: irregular triangle density (a random
    40 % of the base triangles are split into four, a random 30 % of those children again), a few percent of slivers (one
    vertex pulled most of the way to the opposite edge's midpoint), vertex noise of a fraction of the local edge length, and
    a random triangle order in the file.  ~ntris_target triangles, one mesh, deterministic.  Used by `bench.py --standin
    irregular` to check that the numbers quoted on the regular stand-in carry over."""
    rng = np.random.RandomState(seed + 1)
    base = make_dragon(max(1000, int(ntris_target / 3.64)), seed)  # 0.6 + 0.4 * 4 * (0.7 + 0.3 * 4) = 3.64 triangles per base triangle
    P = base.pos_nrm[:, 0:3].astype(np.float64).reshape(-1, 3, 3)
    N = base.pos_nrm[:, 3:6].astype(np.float64).reshape(-1, 3, 3)

    def split(P, N, mask):
        keepP, keepN = P[~mask], N[~mask]
        a, b, c = P[mask][:, 0], P[mask][:, 1], P[mask][:, 2]
        na, nb, nc = N[mask][:, 0], N[mask][:, 1], N[mask][:, 2]
        ab, bc, ca = (a + b) / 2, (b + c) / 2, (c + a) / 2
        nab, nbc, nca = na + nb, nb + nc, nc + na
        kids = [(a, ab, ca, na, nab, nca), (ab, b, bc, nab, nb, nbc), (ca, bc, c, nca, nbc, nc), (ab, bc, ca, nab, nbc, nca)]
        cp = np.concatenate([np.stack(k[0:3], 1) for k in kids])
        cn = np.concatenate([np.stack(k[3:6], 1) for k in kids])
        cn /= np.maximum(np.linalg.norm(cn, axis=-1, keepdims=True), 1e-30)
        return np.concatenate([keepP, cp]), np.concatenate([keepN, cn]), len(keepP)

    P, N, nk = split(P, N, rng.uniform(size=len(P)) < 0.4)
    m2 = np.zeros(len(P), bool)
    m2[nk:] = rng.uniform(size=len(P) - nk) < 0.3
    P, N, _ = split(P, N, m2)
    edge = np.linalg.norm(P[:, 1] - P[:, 0], axis=-1)
    P += rng.normal(size=P.shape) * (0.04 * edge)[:, None, None]  # (per corner: the mesh stays a triangle soup, like assimp's output)
    sl = rng.uniform(size=len(P)) < 0.03
    P[sl, 2] = P[sl, 2] + 0.97 * ((P[sl, 0] + P[sl, 1]) / 2 - P[sl, 2])
    order = rng.permutation(len(P))
    P, N = P[order], N[order]
    ntri = len(P)
    sd = SceneData(
        pos_nrm=np.concatenate([P.reshape(-1, 3), N.reshape(-1, 3)], 1).astype(F32),
        tri=np.arange(3 * ntri, dtype=np.uint32).reshape(ntri, 3),
        tri_mesh=np.zeros(ntri, np.uint32),
        materials=base.materials.copy(),
        point_lights=base.point_lights.copy(),
        name=f"dragon_irregular{ntri}",
    )
    center_and_scale_to_unit(sd)
    return sd


def make_blob(ntris_target: int = 2000, seed: int = 7) -> SceneData:
    """Small seeded procedural multi-mesh scene for fixtures: a bumpy sphere split into 3 meshes."""
    rng = np.random.RandomState(seed)
    nv = max(4, int(round(math.sqrt(ntris_target / 4.0))))
    nu = 2 * nv
    th = np.linspace(0.0, np.pi, nv + 1)
    ph = np.linspace(0.0, 2.0 * np.pi, nu, endpoint=False)
    amp = rng.uniform(0.02, 0.08, size=4)
    fr = rng.randint(2, 9, size=(4, 2))
    TH, PH = np.meshgrid(th, ph, indexing="ij")
    r = 1.0 + sum(a * np.sin(f[0] * TH) * np.cos(f[1] * PH) for a, f in zip(amp, fr))
    P = np.stack([r * np.sin(TH) * np.cos(PH), r * np.cos(TH), r * np.sin(TH) * np.sin(PH)], -1)
    Nn = P / np.maximum(np.linalg.norm(P, axis=-1, keepdims=True), 1e-30)
    rows, tris, tm = [], [], []
    for i in range(nv):
        for j in range(nu):
            j1 = (j + 1) % nu
            quad = [(i, j), (i + 1, j), (i + 1, j1), (i, j1)]
            for a, b, c in ((0, 1, 2), (0, 2, 3)):
                pa, pb, pc = (P[quad[k]] for k in (a, b, c))
                if np.allclose(pa, pb) or np.allclose(pb, pc) or np.allclose(pa, pc):
                    continue  # pole degenerates
                base = len(rows)
                for k in (a, b, c):
                    rows.append(np.concatenate([P[quad[k]], Nn[quad[k]]]))
                tris.append((base, base + 1, base + 2))
                tm.append(0 if i < nv // 3 else (1 if i < 2 * nv // 3 else 2))
    order = np.argsort(np.asarray(tm), kind="stable")
    tri = np.asarray(tris, np.uint32)[order]
    sd = SceneData(
        pos_nrm=np.asarray(rows, F32),
        tri=tri,
        tri_mesh=np.asarray(tm, np.uint32)[order],
        materials=np.asarray(
            [[0.8, 0.2, 0.2, 0.0, 0.0, 0.0, 10, 1], [0.2, 0.8, 0.2, 0.5, 0.5, 0.5, 50, 1], [0.2, 0.2, 0.8, 0.9, 0.9, 0.9, 200, 1]], F32
        ),
        point_lights=np.asarray([[-1, 1, -1, 1, 1, 1], [2, 2, -2, 0.5, 0.5, 0.5]], F32),
        name="blob",
    )
    center_and_scale_to_unit(sd)
    return sd
