"""OBJ ingest (plumbing for running the path without Godot): the reference's Cornell cube written as OBJ text
(numbers from project/demo/geometry/cornell.obj:5-40) loads to the same mesh scenes.cornell_cube_mesh() builds."""
import numpy as np

from gdpathtracing_amd import objio, scenes

CUBE = """
o Cube
v 5 5 -5
v 5 -5 -5
v 5 5 5
v 5 -5 5
v -5 5 -5
v -5 -5 -5
v -5 5 5
v -5 -5 5
vn 0 -1 0
vn 0 0 -1
vn 1 0 0
vn 0 1 0
vn 0 0 1
vt 0.625 0.5
vt 0.375 0.5
usemtl 1
f 1/1/1 3/1/1 7/1/1 5/1/1
f 8/2/3 6/2/3 5/2/3 7/2/3
f 6/1/4 8/1/4 4/1/4 2/1/4
usemtl 2
f 6/1/5 2/1/5 1/1/5 5/1/5
usemtl 3
f 4/1/2 8/1/2 7/1/2 3/1/2
"""


def _tris(mesh):
    out = []
    for s in mesh.surfaces:
        for i in range(0, len(s.indices), 3):
            out.append(tuple(map(tuple, s.vertices[s.indices[i:i + 3]])) + tuple(map(tuple, s.normals[s.indices[i:i + 3]])))
    return out


def test_obj_cube_matches_the_builtin_cornell_cube():
    m = objio.load_obj(CUBE)
    ref = scenes.cornell_cube_mesh()
    assert [len(s.indices) for s in m.surfaces] == [18, 6, 6]
    assert _tris(m) == _tris(ref)          # same triangles, same winding, same normals, same surface split
    # clockwise front faces: cross(e1, e2) opposes the vertex normal
    for s in m.surfaces:
        v = s.vertices[s.indices].reshape(-1, 3, 3)
        n = s.normals[s.indices].reshape(-1, 3, 3)[:, 0]
        g = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
        assert (np.einsum("ij,ij->i", g, n) < 0).all()


def test_obj_without_normals_or_uvs_gets_face_normals():
    m = objio.load_obj("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    s = m.surfaces[0]
    assert len(s.indices) == 3 and np.allclose(s.normals, [[0, 0, 1]] * 3) and (s.uvs == 0).all()


def test_pack_texture_array():
    from gdpathtracing_amd.objio import pack_texture_array
    assert pack_texture_array([], 16).shape == (1, 16, 16, 4) and not pack_texture_array([], 16).any()
    rng = np.random.RandomState(0)
    same = rng.randint(0, 256, size=(8, 8, 4)).astype(np.uint8)
    grey = rng.randint(0, 256, size=(5, 3)).astype(np.uint8)
    rgb = np.zeros((4, 4, 3), np.uint8); rgb[:, 2:] = (200, 100, 50)
    t = pack_texture_array([same, grey, rgb], 8)
    assert t.shape == (3, 8, 8, 4) and t.dtype == np.uint8
    assert np.array_equal(t[0], same)                                  # already the array's resolution: untouched
    assert (t[1][..., 0] == t[1][..., 1]).all() and (t[1][..., 3] == 255).all()   # grey -> rgb, opaque
    assert tuple(t[2][0, 0]) == (0, 0, 0, 255) and tuple(t[2][0, 7]) == (200, 100, 50, 255)
    assert 0 < t[2][0, 3, 0] < 200 or 0 < t[2][0, 4, 0] < 200          # the edge is filtered, not replicated
    flat = np.full((2, 2, 4), 77, np.uint8)
    assert (pack_texture_array([flat], 16) == 77).all()                # a constant image stays constant
