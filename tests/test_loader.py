"""CPU tests of the OBJ/MTL loader of the C++ host mirror (cg-raytracer_amd/host/src/mesh.cpp, replaces src/mesh.cpp:58-166
without assimp) on small files committed under tests/golden/obj/ (written for these tests, so they exist everywhere -- the
reference's data/ directory does not travel).  Loader parity with assimp itself is unpinned (assimp 5.0.1 is absent from the
image); what is pinned here: the documented semantics (per-face-corner vertices, one mesh per object and material
run, objects in reverse file order, fan triangulation, flat normals when a face has none, relative indices, MTL fields), the
agreement of the C++ loader with the Python loader that made the committed scene fixtures, and the failure behaviour:
an index that is 0, out of range, or not a number makes loadMesh print upstream's message and throw (mesh.cpp:66-71)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

OBJ = os.path.join(GOLDEN, "obj")


def test_small_obj_semantics(pkg):
    sd = pkg.host_load_obj(os.path.join(OBJ, "small.obj"))
    # objects in reverse file order; inside an object one mesh per material run
    assert sd.nmesh == 4 and sd.ntris == 5
    assert sd.tri_mesh.tolist() == [0, 1, 2, 2, 3]
    # mesh 0: `second`, default material (no usemtl yet in that object -> the run keeps the previous material name "mirror")
    mirror = [0.1, 0.1, 0.1, 0.9, 0.9, 0.9, 200.0, 1.0]
    red = [0.8, 0.1, 0.1, 0.2, 0.2, 0.2, 12.0, 1.0]
    assert np.allclose(sd.materials[0], mirror) and np.allclose(sd.materials[1], red)
    assert np.allclose(sd.materials[2], red) and np.allclose(sd.materials[3], mirror)
    # per-face-corner vertices: 3 + 3 + 4 + 3
    assert len(sd.pos_nrm) == 13
    # the face given with negative indices is the triangle (0,0,2) (1,0,2) (0,1,2), normals generated: flat +z
    assert np.array_equal(sd.pos_nrm[0:3, 0:3], np.float32([[0, 0, 2], [1, 0, 2], [0, 1, 2]]))
    assert np.allclose(sd.pos_nrm[0:3, 3:6], [0, 0, 1])
    # the quad is fan-triangulated from its first corner
    quad = sd.tri[sd.tri_mesh == 2]
    assert (quad - quad.min()).tolist() == [[0, 1, 2], [0, 2, 3]]
    # `f 1 2 3` has no normals: generated flat normal
    assert np.allclose(sd.pos_nrm[sd.tri[4], 3:6], [0, 0, 1])


def test_cpp_loader_equals_python_loader(pkg):
    for normalize in (False, True):
        a = pkg.host_load_obj(os.path.join(OBJ, "small.obj"), normalize=normalize)
        b = pkg.scenes.load_obj(os.path.join(OBJ, "small.obj"), normalize=normalize)
        assert np.array_equal(a.tri, b.tri) and np.array_equal(a.tri_mesh, b.tri_mesh)
        assert a.pos_nrm.tobytes() == b.pos_nrm.tobytes() and a.materials.tobytes() == b.materials.tobytes()
    n = pkg.host_load_obj(os.path.join(OBJ, "small.obj"), normalize=True)
    assert np.abs(np.linalg.norm(n.pos_nrm[:, 0:3], axis=1).max() - 1.0) < 1e-6  # centred, farthest vertex at distance 1 (mesh.cpp:143-166)


@pytest.mark.parametrize("name", ["bad_out_of_range", "bad_zero_index", "bad_negative", "bad_token", "bad_normal_index", "does_not_exist"])
def test_invalid_files_throw_like_upstream(pkg, name, capfd):
    """Malformed faces are REJECTED with upstream's message and a bare exception (mesh.cpp:60-71).  PARITY UNPINNED: which malformed
    faces assimp 5.0.1 refuses rather than repairs or skips is recorded nowhere in the reference and assimp is not importable
    here; the bad_*.obj fixtures are this repo's own expectation (an index that cannot be resolved must not become a triangle),
    not a statement about upstream's rejected set."""
    with pytest.raises(RuntimeError):
        pkg.host_load_obj(os.path.join(OBJ, name + ".obj"))
    err = capfd.readouterr().err
    assert ("does not exist" in err) if name == "does_not_exist" else ("Assimp failed to load mesh file" in err)
