import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    p = entry.load_package()
    if not (os.path.exists(p.LIB_PATH) and os.path.exists(p.HOST_LIB_PATH)):
        p.build_native()
    return p


@pytest.fixture(scope="session")
def orc():
    o = entry.load_oracle()
    if not os.path.exists(o.LIB):
        o.build()
    return o


@pytest.fixture(scope="session")
def scene_data(pkg):
    """name -> SceneData from the committed fixtures (tests/golden/scenes/*.npz)."""
    cache = {}

    def get(name):
        if name not in cache:
            if name == "blob":
                cache[name] = pkg.scenes.make_blob(2000, seed=7)
            elif name == "spheres":
                cache[name] = pkg.scenes.spheres_preset()
            else:
                cache[name] = pkg.scenes.SceneData.load(os.path.join(GOLDEN, "scenes", name + ".npz"))
                cache[name].name = name
        return cache[name]

    return get


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bits(a, b):
    """Bit-pattern equality of float32 arrays; NaNs compare equal to NaNs (x86 SSE produces the negative
    default NaN 0xffc00000 for 0/0, gfx950 the positive 0x7fc00000 -- the payload is not part of any result
    the reference can observe, every comparison with a NaN being false either way)."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
