"""Seeded ray families for parity tests: a primary grid plus adversarial rays --
axis-parallel directions (zero components => +-inf / NaN slabs), origins on box faces, inside boxes and on
triangle planes, finite ray.t (shadow style), edge- and vertex-grazing rays, and the three cube.obj rays on
which the reference's BVH misses a hit its brute force finds."""
import numpy as np

FMAX = np.finfo(np.float32).max
SEED = 0x5EEDC0DE  # recorded in every fixture header

F4_ORIGIN = (1.1, 1.3, -2.6)
F4_DIRS = [
    (-0.355097264, -0.411639154, 0.839320719),
    (-0.378721297, -0.235073969, 0.895159364),
    (-0.380845577, -0.211262539, 0.900180399),
]
# what the reference itself produced for them during the survey probe:
F4_BRUTE_T = ["3.09774303", "2.90451074", "2.88830972"]


def _mk(o, d, t=None):
    o = np.asarray(o, np.float32).reshape(-1, 3)
    d = np.asarray(d, np.float32).reshape(-1, 3)
    n = max(len(o), len(d))
    r = np.zeros((n, 7), np.float32)
    r[:, 0:3] = o
    r[:, 3:6] = d
    r[:, 6] = FMAX if t is None else np.asarray(t, np.float32)
    return r


def _norm(v):
    v = np.asarray(v, np.float64)
    return (v / np.maximum(np.linalg.norm(v, axis=-1, keepdims=True), 1e-30)).astype(np.float32)


def families(sd, node_boxes, primary, rng=None, n_random=1500):
    """sd: SceneData; node_boxes: (N,6) AABBs of the reference tree; primary: (P,7) camera rays.
    Returns dict name -> (n,7) float32 rays."""
    rng = np.random.RandomState(SEED & 0x7FFFFFFF) if rng is None else rng
    fam = {"primary": primary.astype(np.float32)}
    cam_pos = primary[0, 0:3] if len(primary) else np.array([1.0, 1.0, -2.5], np.float32)

    # random origins / directions, normalised and not
    o = rng.uniform(-2.0, 2.0, (n_random, 3))
    tgt = rng.uniform(-0.6, 0.6, (n_random, 3))
    d = tgt - o
    fam["random_unit"] = _mk(o, _norm(d))
    fam["random_scaled"] = _mk(o, (d * rng.uniform(0.05, 7.0, (n_random, 1))).astype(np.float32))
    # finite ray.t (shadow-style), incl. t shorter than the hit distance, zero and tiny
    tt = rng.uniform(0.0, 4.0, n_random).astype(np.float32)
    tt[:16] = [0.0, 1e-30, 1e-6, 1e-3, 0.5, 1.0, 2.0, 3.0, 1e10, FMAX, np.inf, 0.25, 0.75, 1.5, 2.5, 3.5]
    fam["finite_t"] = _mk(o, _norm(d), tt)

    V = sd.pos_nrm[:, 0:3]
    T = sd.tri
    if len(T):
        k = rng.randint(0, len(T), min(600, 40 * len(T)))
        A, B, C = V[T[k, 0]], V[T[k, 1]], V[T[k, 2]]
        # vertex- and edge-grazing from the camera position, and from random origins
        for name, P in (("graze_vertex", A), ("graze_edge", ((A.astype(np.float64) + B) / 2).astype(np.float32)),
                        ("graze_centroid", ((A.astype(np.float64) + B + C) / 3).astype(np.float32))):
            fam[name] = _mk(np.broadcast_to(cam_pos, P.shape), _norm(P - cam_pos))
        # origins ON triangle planes (vertex, edge midpoint, interior point) shooting along +-normal and sideways
        nrm = np.cross(B - A, C - A)
        w = rng.dirichlet((1, 1, 1), len(k)).astype(np.float32)
        inner = (w[:, 0:1] * A + w[:, 1:2] * B + w[:, 2:3] * C).astype(np.float32)
        oo = np.concatenate([A, ((A + B) * np.float32(0.5)).astype(np.float32), inner, inner])
        dd = np.concatenate([_norm(nrm), _norm(-nrm), _norm(nrm), _norm(B - A)])
        fam["on_plane"] = _mk(oo, dd)

    if len(node_boxes):
        bx = node_boxes[rng.randint(0, len(node_boxes), 400)]
        lo, hi = bx[:, 0:3], bx[:, 3:6]
        ctr = ((lo.astype(np.float64) + hi) / 2).astype(np.float32)
        axes = np.eye(3, dtype=np.float32)
        oo, dd = [], []
        for a in range(3):
            for sgn in (1.0, -1.0):
                # axis-parallel rays: from outside through the box centre, from ON the lower/upper face, from the centre
                dvec = axes[a] * np.float32(sgn)
                start = ctr.copy()
                start[:, a] = lo[:, a] - 1.5 if sgn > 0 else hi[:, a] + 1.5
                oo += [start, lo, hi, ctr]
                dd += [np.broadcast_to(dvec, ctr.shape)] * 4
                # two zero components and an origin exactly in a face plane but outside the face
                off = lo.copy()
                off[:, (a + 1) % 3] -= 0.25
                oo.append(off)
                dd.append(np.broadcast_to(dvec, ctr.shape))
        # one zero component (diagonal in a plane), origins at box corners / centres
        for a in range(3):
            dvec = np.ones(3, np.float32)
            dvec[a] = 0.0
            oo += [lo - np.float32(0.5) * dvec, ctr]
            dd += [np.broadcast_to(_norm(dvec), ctr.shape)] * 2
        fam["axis_parallel"] = _mk(np.concatenate(oo), np.concatenate(dd))
        # inside boxes with random directions, and negative-zero direction components
        dz = _norm(rng.normal(size=(len(ctr), 3)))
        dz[::3, 0] = -0.0
        dz[1::3, 1] = 0.0
        fam["inside_boxes"] = _mk(ctr, dz)

    # extreme magnitudes: far-away origins aimed at the scene, tiny / huge direction scales, mixed
    # (exercise the "irregular ray" fallback of the in-leaf accelerator and overflow/underflow paths)
    m = 240
    tgt2 = rng.uniform(-0.5, 0.5, (m, 3))
    dirs = _norm(rng.normal(size=(m, 3))).astype(np.float64)
    dist = np.repeat([1e3, 1e6, 1e9, 1e13, 3e-3, 30.0], m // 6)[:, None]
    oo = (tgt2 - dirs * dist).astype(np.float32)
    dd = dirs.astype(np.float32)
    scl = np.tile(np.float32([1.0, 1e-15, 1e15, 1e-30, 1e30, 1e-38]), m // 6)[:, None]
    fam["extreme"] = _mk(oo, (dd * scl).astype(np.float32))

    fam["f4_cube"] = _mk(np.broadcast_to(np.float32(F4_ORIGIN), (3, 3)), np.float32(F4_DIRS))
    return fam


def concat(fam):
    return np.concatenate([fam[k] for k in sorted(fam)]).astype(np.float32)
