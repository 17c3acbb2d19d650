"""The lemma behind the certified walk's reduced certificate (walk_fast.h path_certified), checked
with the oracle's restatement of the reference's box arithmetic (ray_tracing.cpp:162-200, bvh.cpp:647-661) on the CPU:

    inside the no-NaN envelope (finite operands, no zero direction component), if a box B is "entered at t" --
    origin strictly inside B, or the slab test passes (no `cur >= ray.t` rejection) with its parameter cur <= t --
    then so is every box A that contains B (A.lo <= B.lo and A.hi >= B.hi, componentwise, as floats).

That is why a certificate only tests the path boxes that do NOT contain their successor (outside the envelope NaNs enter the
ternary ladders and path_certified keeps testing every box: nothing is claimed there).  Random and adversarial cases:
origins on faces and corners of either box, shared faces, zero-thickness boxes, tiny and huge direction components, t = 0."""
import numpy as np
import pytest


def _entered(orc, boxes, rays, t):
    r = rays.copy()
    r[:, 6] = np.inf  # no `cur >= ray.t` rejection: the geometric part of the test and its parameter
    out = orc.ray_box(boxes, r)
    inside = out["pad"] != 0  # oracle_ray_box reports startsInBox in the record's last word
    hit = out["hit"] == 1
    cur = out["t"]
    with np.errstate(invalid="ignore"):
        return inside | (hit & ~(cur > t))


def _cases(rng, n):
    inner_lo = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    ext = rng.uniform(0, 1.5, (n, 3)).astype(np.float32)
    ext[rng.rand(n, 3) < 0.08] = 0.0  # zero-thickness boxes (axis-aligned faces)
    inner_hi = (inner_lo + ext).astype(np.float32)
    grow_lo = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    grow_hi = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    grow_lo[rng.rand(n, 3) < 0.3] = 0.0  # shared faces
    grow_hi[rng.rand(n, 3) < 0.3] = 0.0
    outer_lo = np.minimum((inner_lo - grow_lo).astype(np.float32), inner_lo)
    outer_hi = np.maximum((inner_hi + grow_hi).astype(np.float32), inner_hi)
    o = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
    # origins snapped onto planes of either box, per axis
    for planes in (inner_lo, inner_hi, outer_lo, outer_hi):
        m = rng.rand(n, 3) < 0.07
        o[m] = planes[m]
    d = rng.normal(size=(n, 3)).astype(np.float32)
    aim = rng.rand(n) < 0.6  # most rays are aimed at a point of the inner box, so that entering is the common outcome
    target = (inner_lo + rng.rand(n, 3).astype(np.float32) * (inner_hi - inner_lo)).astype(np.float32)
    d[aim] = (target[aim] - o[aim]).astype(np.float32)
    scale = np.float32(2.0) ** rng.randint(-30, 31, (n, 3)).astype(np.float32)
    wild = rng.rand(n, 3) < 0.1
    d[wild] = (d[wild] * scale[wild]).astype(np.float32)
    d[np.abs(d) < 2.0 ** -36] = np.float32(2.0 ** -36)  # the envelope: no zero component (RayFast::fd)
    rays = np.concatenate([o, d, np.zeros((n, 1), np.float32)], 1).astype(np.float32)
    t = rng.uniform(0, 2.5, n).astype(np.float32)  # (an aimed ray reaches its target at parameter 1)
    t[rng.rand(n) < 0.05] = 0.0
    return (np.concatenate([inner_lo, inner_hi], 1), np.concatenate([outer_lo, outer_hi], 1), rays, t)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_a_box_containing_an_entered_box_is_entered(orc, seed):
    rng = np.random.RandomState(seed)
    inner, outer, rays, t = _cases(rng, 400_000)
    assert (outer[:, 0:3] <= inner[:, 0:3]).all() and (outer[:, 3:6] >= inner[:, 3:6]).all()
    e_in = _entered(orc, inner, rays, t)
    # t at the inner box's own parameter: the tightest case (cur_inner == t exactly)
    r = rays.copy(); r[:, 6] = np.inf
    cur = orc.ray_box(inner, r)["t"]
    e_out = _entered(orc, outer, rays, t)
    bad = e_in & ~e_out
    assert not bad.any(), f"{bad.sum()} of {e_in.sum()} entered inner boxes with an outer box that is not entered; first: {np.flatnonzero(bad)[:3]}"
    hit = orc.ray_box(inner, r)["hit"] == 1
    tt = np.where(hit, cur, t).astype(np.float32)
    e_in2 = _entered(orc, inner, rays, tt)
    e_out2 = _entered(orc, outer, rays, tt)
    assert e_in2[hit].all()  # a box is entered at its own parameter
    bad2 = e_in2 & ~e_out2
    assert not bad2.any(), f"{bad2.sum()} cases at cur == t"
    assert e_in.sum() > 20_000 and (~e_in).sum() > 20_000  # both outcomes are exercised
