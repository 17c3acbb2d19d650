"""Time-boxed fuzz in the driver-run suite.  The certified walk returns the reference's answer by PROOF
(walk_fast.h); the evidence for that proof must not live only in builder-run logs, so tools/fuzz_parity.py and
tools/fuzz_render.py run here as library calls, ~45 s each, split over the committed seed list (tests/fuzz_seeds.txt, to which
every round appends one seed):
  * parity: random scenes (soups with degenerate / duplicated / axis-aligned triangles, slivers, dragon stand-ins, blobs, scales
    2^-20 .. 2^37) x ray families (tests/rayfam.py + vertex/edge-aimed rays); the default walk in BOTH kernel shapes, the exact walk
    and the device brute force against the oracle on flag, t bits, primitive id, material id and normal bits; and primary frames
    with frame hints (hard tiles first / as 16-ray waves, thresholds of 20 .. 400 ticks) against the plain frame byte for byte;
  * render: cgrt_render (random scene, camera, 1-3 point lights, depth 0-4) against the oracle's recursive per-pixel driver:
    RGB <= 1e-5 (BASELINE.json north_star's tolerance), equal ray counts, certified == exact == quad-shape frames byte for byte
    (the first frame of a scene is exactly sized, the later ones are predicted frames: capi.cpp render_impl).
The fallback-ray count is asserted > 0 so that the path "no certificate -> exact walk" stays exercised."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu
BUDGET_S = float(os.environ.get("CGRT_FUZZ_SECONDS", "45"))


def _seeds():
    out = []
    for line in open(os.path.join(ROOT, "tests", "fuzz_seeds.txt")):
        line = line.split("#")[0].split()
        if len(line) == 3:
            out.append((line[0], int(line[1]), int(line[2])))
    assert len(out) >= 3
    return out


def test_fuzz_parity_time_boxed(pkg, orc):
    import fuzz_parity

    tot = {}
    seeds = _seeds()
    for _, ps, _ in seeds:
        st = fuzz_parity.run(BUDGET_S / len(seeds), ps, verbose=False)
        for k, v in st.items():
            tot[k] = tot.get(k, 0) + v
    print("fuzz parity:", tot)
    assert tot["mismatching_batches"] == 0, tot
    assert tot["scenes"] >= len(seeds) and tot["rays"] > 10_000
    assert tot["certified_scenes"] > 0 and tot["quad_shape_batches"] > 0 and tot["tree_rays"] > 0 and tot["hinted_frames"] > 0
    assert tot["fallback_rays"] > 0, "the no-certificate -> exact-walk path must stay exercised"


def test_fuzz_render_time_boxed(pkg, orc):
    import fuzz_render

    tot = {}
    seeds = _seeds()
    for _, _, rs in seeds:
        st = fuzz_render.run(BUDGET_S / len(seeds), rs, verbose=False)
        for k, v in st.items():
            tot[k] = max(tot.get(k, 0.0), v) if k == "max_err" else tot.get(k, 0) + v
    print("fuzz render:", tot)
    assert tot["bad_rgb"] == 0 and tot["bad_counts"] == 0 and tot["walks_differ"] == 0 and tot["shapes_differ"] == 0, tot
    assert tot["max_err"] <= 1e-5  # BASELINE.json north_star: final pixel RGB within 1e-5 abs
    assert tot["frames"] >= len(seeds) and tot["certified_frames"] > 0
