"""Synthetic scenes and camera blocks for tests and bench.

No reference asset travels: everything here is generated from seeds and from the
public numbers in project/demo/demo.tscn:19-93 and project/demo/geometry/cornell.obj:5-40
(transforms, material colours, box dimensions).

A scene is described the way the reference's GeometryGroup3D sees it
(src/path_tracing/geometry_group3d.cpp:228-366): unique meshes made of surfaces
(vertex / normal / uv / index arrays, bvh.cpp:192-198), instances (mesh id, Transform3D,
up to three material ids) and a material table whose entry 0 is the default material.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import wire


@dataclass
class Surface:
    vertices: np.ndarray  # (n,3) f32
    normals: np.ndarray   # (n,3) f32
    uvs: np.ndarray       # (n,2) f32
    indices: np.ndarray   # (m,) i32, m % 3 == 0

    def __post_init__(self):
        self.vertices = np.ascontiguousarray(self.vertices, dtype=np.float32).reshape(-1, 3)
        self.normals = np.ascontiguousarray(self.normals, dtype=np.float32).reshape(-1, 3)
        self.uvs = np.ascontiguousarray(self.uvs, dtype=np.float32).reshape(-1, 2)
        self.indices = np.ascontiguousarray(self.indices, dtype=np.int32).reshape(-1)


@dataclass
class Mesh:
    surfaces: List[Surface]

    @property
    def n_tris(self) -> int:
        return sum(len(s.indices) // 3 for s in self.surfaces)


@dataclass
class Instance:
    mesh: int
    transform: np.ndarray            # (12,) f32: basis rows (xx xy xz yx yy yz zx zy zz) + origin
    material_ids: List[int] = field(default_factory=lambda: [0])


@dataclass
class CameraDesc:
    transform: np.ndarray            # (12,) like Instance.transform
    fov_deg: float = 79.5            # demo.tscn:50
    near: float = 0.01               # path_tracing_camera.cpp:134
    far: float = 1000.0


@dataclass
class Scene:
    name: str
    meshes: List[Mesh]
    instances: List[Instance]
    materials: np.ndarray            # wire.MATERIAL[]
    camera: CameraDesc
    textures: Optional[np.ndarray] = None  # (layers,res,res,4) u8

    @property
    def n_unique_tris(self) -> int:
        return sum(m.n_tris for m in self.meshes)

    @property
    def n_instanced_tris(self) -> int:
        return sum(self.meshes[i.mesh].n_tris for i in self.instances)


# --------------------------------------------------------------------------- helpers

def transform12(basis_rows=None, origin=(0.0, 0.0, 0.0)) -> np.ndarray:
    b = np.eye(3) if basis_rows is None else np.asarray(basis_rows, dtype=np.float64).reshape(3, 3)
    return np.concatenate([b.reshape(-1), np.asarray(origin, dtype=np.float64)]).astype(np.float32)


def rot_y(deg: float) -> np.ndarray:
    a = np.deg2rad(deg)
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def material(albedo=(1.0, 1.0, 1.0), emission=(0.0, 0.0, 0.0), energy=1.0, metallic=0.0, roughness=1.0,
             texture=-1) -> np.ndarray:
    """StandardMaterial3D -> GpuMaterial (geometry_group3d.cpp:279-290)."""
    m = np.zeros((), dtype=wire.MATERIAL)
    m["albedo"] = (*albedo, 1.0)
    m["emission"] = (*emission, energy)
    m["metallic"] = metallic
    m["roughness"] = roughness
    m["albedo_texture_index"] = texture
    return m


def _quad(p0, p1, p2, p3, normal, uv=((0, 0), (1, 0), (1, 1), (0, 1))):
    """Corners in counter-clockwise order seen from the normal side (OBJ convention); emitted with Godot's
    clockwise winding, i.e. cross(e1, e2) opposes the vertex normal (main.glsl:254-255, :208 rely on it)."""
    v = np.array([p0, p1, p2, p3], dtype=np.float32)
    n = np.tile(np.asarray(normal, dtype=np.float32), (4, 1))
    return v, n, np.asarray(uv, dtype=np.float32), np.array([0, 2, 1, 0, 3, 2], dtype=np.int32)


def _merge(quads) -> Surface:
    vs, ns, us, idx, base = [], [], [], [], 0
    for v, n, u, i in quads:
        vs.append(v); ns.append(n); us.append(u); idx.append(i + base)
        base += len(v)
    return Surface(np.concatenate(vs), np.concatenate(ns), np.concatenate(us), np.concatenate(idx))


def plane_mesh(size=2.0) -> Mesh:
    """Godot PlaneMesh default (2x2, +Y up), 2 triangles (demo.tscn:19)."""
    h = size / 2
    return Mesh([_merge([_quad((-h, 0, -h), (-h, 0, h), (h, 0, h), (h, 0, -h), (0, 1, 0))])])


def cornell_cube_mesh() -> Mesh:
    """Open 10x10x10 cube, 5 quads in 3 surfaces, normals pointing inwards (cornell.obj:5-40):
    surface 0 = top (y=+5), x=-5 wall, bottom (y=-5); surface 1 = z=-5 wall; surface 2 = z=+5 wall;
    the x=+5 side is open."""
    v = {1: (5, 5, -5), 2: (5, -5, -5), 3: (5, 5, 5), 4: (5, -5, 5), 5: (-5, 5, -5), 6: (-5, -5, -5),
         7: (-5, 5, 5), 8: (-5, -5, 5)}
    s0 = _merge([
        _quad(v[1], v[3], v[7], v[5], (0, -1, 0), ((0.625, 0.5), (0.625, 0.75), (0.875, 0.75), (0.875, 0.5))),
        _quad(v[8], v[6], v[5], v[7], (1, 0, 0), ((0.375, 0.0), (0.375, 0.25), (0.625, 0.25), (0.625, 0.0))),
        _quad(v[6], v[8], v[4], v[2], (0, 1, 0), ((0.125, 0.5), (0.125, 0.75), (0.375, 0.75), (0.375, 0.5))),
    ])
    s1 = _merge([_quad(v[6], v[2], v[1], v[5], (0, 0, 1), ((0.375, 0.25), (0.375, 0.5), (0.625, 0.5), (0.625, 0.25)))])
    s2 = _merge([_quad(v[4], v[8], v[7], v[3], (0, 0, -1), ((0.375, 0.75), (0.375, 1.0), (0.625, 1.0), (0.625, 0.75)))])
    return Mesh([s0, s1, s2])


def box_mesh(sx, sy, sz) -> Mesh:
    """Closed axis-aligned box centred at the origin, outward normals, 12 triangles."""
    x, y, z = sx / 2, sy / 2, sz / 2
    q = [
        _quad((-x, -y, z), (x, -y, z), (x, y, z), (-x, y, z), (0, 0, 1)),
        _quad((x, -y, -z), (-x, -y, -z), (-x, y, -z), (x, y, -z), (0, 0, -1)),
        _quad((x, -y, z), (x, -y, -z), (x, y, -z), (x, y, z), (1, 0, 0)),
        _quad((-x, -y, -z), (-x, -y, z), (-x, y, z), (-x, y, -z), (-1, 0, 0)),
        _quad((-x, y, z), (x, y, z), (x, y, -z), (-x, y, -z), (0, 1, 0)),
        _quad((-x, -y, -z), (x, -y, -z), (x, -y, z), (-x, -y, z), (0, -1, 0)),
    ]
    return Mesh([_merge(q)])


def blob_mesh(n_tris=51200, seed=1, major=0.55, minor=0.32) -> Mesh:
    """Procedural 'character' stand-in: a displaced torus, nu*nv quads -> 2*nu*nv triangles.
    Deterministic from `seed`; normals from the displaced surface."""
    nq = n_tris // 2
    nu = int(round(np.sqrt(nq)))
    while nq % nu:
        nu -= 1
    nv = nq // nu
    rng = np.random.RandomState(seed)
    ph = rng.uniform(0, 2 * np.pi, size=6)
    u = (np.arange(nu) / nu)[:, None] * 2 * np.pi
    v = (np.arange(nv) / nv)[None, :] * 2 * np.pi

    def surf(u, v):
        d = (0.060 * np.sin(5 * u + ph[0]) * np.sin(3 * v + ph[1]) + 0.030 * np.sin(11 * u + 7 * v + ph[2])
             + 0.015 * np.sin(23 * u - 17 * v + ph[3]) + 0.05 * np.sin(2 * u + ph[4]) * np.cos(v + ph[5]))
        r = minor + d
        x = (major + r * np.cos(v)) * np.cos(u)
        z = (major + r * np.cos(v)) * np.sin(u)
        y = r * np.sin(v) * 1.25
        return np.stack([x, y, z], axis=-1)

    p = surf(u, v)
    e = 1e-4
    du = surf(u + e, v) - surf(u - e, v)
    dv = surf(u, v + e) - surf(u, v - e)
    n = np.cross(dv, du)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    uu = np.broadcast_to(np.arange(nu)[:, None] / nu, (nu, nv))
    vv = np.broadcast_to(np.arange(nv)[None, :] / nv, (nu, nv))
    uv = np.stack([uu, vv], axis=-1)
    i0 = (np.arange(nu)[:, None] * nv + np.arange(nv)[None, :])
    i1 = (((np.arange(nu) + 1) % nu)[:, None] * nv + np.arange(nv)[None, :])
    j1 = (np.arange(nv) + 1) % nv
    a = i0
    b = i1
    c = (((np.arange(nu) + 1) % nu)[:, None] * nv + j1[None, :])
    d = (np.arange(nu)[:, None] * nv + j1[None, :])
    idx = np.stack([a, b, c, a, c, d], axis=-1).reshape(-1)
    return Mesh([Surface(p.reshape(-1, 3), n.reshape(-1, 3), uv.reshape(-1, 2), idx)])


# demo.tscn material table (demo.tscn:21-45); index 0 = default StandardMaterial3D
def _demo_materials() -> np.ndarray:
    return np.stack([
        material(),                                                                     # 0 default
        material(emission=(0.832472, 0.8072, 0.719802), energy=10.0),                   # 1 light   :21-24
        material(roughness=0.6),                                                        # 2 white   :26-27
        material(albedo=(1.0, 0.16, 0.16)),                                             # 3 red     :29-30
        material(albedo=(0.42, 1.0, 0.13)),                                             # 4 green   :32-33
        material(albedo=(0.8, 0.8, 0.8), emission=(0.360742, 0.135649, 0.818479), energy=0.4),  # 5 :35-39
        material(metallic=1.0, roughness=0.16),                                         # 6 metal   :41-43
    ])


_LIGHT_T = transform12([[1, 0, 0], [0, -1, 1.50996e-07], [0, -1.50996e-07, -1]], (0, 2.95581, 0))       # demo.tscn:74
_BOX_T = transform12([[-2.62268e-08, 0, -0.6], [0, 0.6, 0], [0.6, 0, -2.62268e-08]], (0, 0, 0))          # demo.tscn:79
_CHAR1_T = transform12([[0.982635, -0.208021, 0.656626], [0.0853118, 1.17191, 0.243597],
                        [-0.68348, -0.152791, 0.974428]], (-1.16402, -1.55573, -0.923088))               # demo.tscn:86
_CHAR2_T = transform12([[0.934979, 0.0872355, -0.747128], [0.0853118, 1.17191, 0.243597],
                        [0.74735, -0.242915, 0.906899]], (1.27032, -0.951083, -0.923088))                # demo.tscn:91
def _demo_camera() -> CameraDesc:   # demo.tscn:50-53; a new object per scene so a test may move it
    return CameraDesc(transform12(None, (0, 0, 9.7694)))


def cornell_scene() -> Scene:
    """S-cornell (config C1): open cube + light + short and tall blocks = 36 triangles, 4 instances."""
    meshes = [plane_mesh(), cornell_cube_mesh(), box_mesh(1.7, 1.7, 1.7), box_mesh(1.7, 3.4, 1.7)]
    inst = [
        Instance(0, _LIGHT_T, [1]),
        Instance(1, _BOX_T, [2, 3, 4]),
        Instance(2, transform12(rot_y(-17.0), (1.0, -3.0 + 0.85, 0.9)), [2]),
        Instance(3, transform12(rot_y(20.0), (-1.0, -3.0 + 1.7, -0.9)), [2]),
    ]
    return Scene("cornell", meshes, inst, _demo_materials(), _demo_camera())


def demo_scene(n_tris=51200, seed=1) -> Scene:
    """S-demo (configs C2/C3/C5): demo.tscn's composition -- light, open cube and two instances of one
    'character' mesh (emissive-tinted and metallic), the character being a procedural blob of n_tris
    triangles in place of the Gobot/Suzanne assets."""
    meshes = [plane_mesh(), cornell_cube_mesh(), blob_mesh(n_tris, seed)]
    inst = [
        Instance(0, _LIGHT_T, [1]),
        Instance(1, _BOX_T, [2, 3, 4]),
        Instance(2, _CHAR1_T, [5]),
        Instance(2, _CHAR2_T, [6]),
    ]
    return Scene("demo%d" % n_tris, meshes, inst, _demo_materials(), _demo_camera())


def instanced_scene(n_side=32, n_unique=8, tris_per_mesh=1024, seed=7) -> Scene:
    """S-inst (config C4): n_side^2 instances of n_unique blob meshes on a jittered grid above a ground
    quad, lit by a large emissive plane -> deep TLAS."""
    rng = np.random.RandomState(seed)
    meshes = [plane_mesh(), plane_mesh()] + [blob_mesh(tris_per_mesh, seed=100 + k) for k in range(n_unique)]
    span = 12.0
    inst = [
        Instance(0, transform12([[4, 0, 0], [0, -1, 0], [0, 0, -4]], (0, 7.0, 0)), [1]),
        Instance(1, transform12(np.eye(3) * span, (0, -1.2, 0)), [2]),
    ]
    cell = 2 * span / n_side
    mats = [3, 4, 5, 6, 2]
    for iz in range(n_side):
        for ix in range(n_side):
            s = 0.30 * cell * rng.uniform(0.8, 1.3)
            b = rot_y(rng.uniform(0, 360)) * s
            o = (-span + (ix + 0.5) * cell + rng.uniform(-0.1, 0.1) * cell, -0.9 + rng.uniform(0, 1.5),
                 -span + (iz + 0.5) * cell + rng.uniform(-0.1, 0.1) * cell)
            inst.append(Instance(2 + rng.randint(n_unique), transform12(b, o), [mats[rng.randint(len(mats))]]))
    cam_basis = np.array([[1, 0, 0], [0, np.cos(0.5), np.sin(0.5)], [0, -np.sin(0.5), np.cos(0.5)]])
    cam = CameraDesc(transform12(cam_basis, (0, 9.0, 15.0)))
    return Scene("inst%d" % (n_side * n_side), meshes, inst, _demo_materials(), cam)


def unique_scene(n_tris=1_000_000, seed=3) -> Scene:
    """S-unique (the stress variant of C4): ONE BLAS of n_tris unique triangles -- a finely displaced
    blob that fills the open cube -- plus the cube and the light, seen from the box opening so that every pixel traces.
    At the default size the reference layout is ~160 MB (48 + 80 B per triangle + nodes) and the flattened records
    ~110 MB: past the 32 MB of L2, inside the 256 MB Infinity Cache; n_tris = 4_000_000 leaves that too.  The only
    configuration where the record layout meets the memory system rather than the caches."""
    meshes = [plane_mesh(), cornell_cube_mesh(), blob_mesh(n_tris, seed, major=0.62, minor=0.30)]
    inst = [
        Instance(0, _LIGHT_T, [1]),
        Instance(1, _BOX_T, [2, 3, 4]),
        Instance(2, transform12(rot_y(25.0) * 3.1, (0.0, -0.9, -0.4)), [5]),
    ]
    cam = CameraDesc(transform12(None, (0.0, 0.0, 4.2)), fov_deg=75.0)
    return Scene("unique%d" % n_tris, meshes, inst, _demo_materials(), cam)


def random_scene(seed: int, n_meshes: int = 4, n_instances: int = 9, tris_per_surface: int = 40, textured: bool = True,
                 coincident: bool = True) -> Scene:
    """Fuzz input: triangle soups (slivers, a few degenerate triangles, non-unit and inconsistent vertex normals,
    uvs outside [0,1]), meshes of 1..3 surfaces, instances with non-uniform scale, shear and mirroring, random
    materials (fractional metallic, roughness 0, emitters, textured and untextured)."""
    rng = np.random.RandomState(seed)
    meshes = []
    for _ in range(n_meshes):
        surfaces = []
        for _s in range(rng.randint(1, 4)):
            n = tris_per_surface
            centres = rng.uniform(-1.0, 1.0, size=(n, 1, 3))
            v = (centres + rng.normal(0.0, 0.35, size=(n, 3, 3))).astype(np.float32)
            v[0, 2] = v[0, 1]                                   # a degenerate triangle (two equal vertices)
            v[1, 2] = v[1, 0] + (v[1, 1] - v[1, 0]) * 0.5       # a zero-area triangle (collinear)
            if coincident:
                v[2] = v[3]                                     # two coincident triangles (equal t on every ray: the
                                                                # winner depends on the visiting order, i.e. on the tree)
            nrm = rng.normal(size=(n, 3, 3)).astype(np.float32) * rng.uniform(0.2, 3.0, size=(n, 1, 1)).astype(np.float32)
            uv = rng.uniform(-0.5, 1.5, size=(n, 3, 2)).astype(np.float32)
            surfaces.append(Surface(v.reshape(-1, 3), nrm.reshape(-1, 3), uv.reshape(-1, 2), np.arange(3 * n, dtype=np.int32)))
        meshes.append(Mesh(surfaces))
    n_mat = 8
    mats = [material()]
    for k in range(1, n_mat):
        emit = rng.rand() < 0.35
        mats.append(material(albedo=tuple(rng.uniform(0.05, 1.0, size=3)),
                             emission=tuple(rng.uniform(0, 1, size=3)) if emit else (0.0, 0.0, 0.0),
                             energy=float(rng.uniform(0.5, 6.0)) if emit else 1.0,
                             metallic=float(rng.choice([0.0, 1.0, rng.rand()])),
                             roughness=float(rng.choice([0.0, 0.003, 0.3, 1.0, rng.rand()])),
                             texture=int(rng.randint(0, 2)) if (textured and rng.rand() < 0.5) else -1))
    inst = []
    for _ in range(n_instances):
        m = int(rng.randint(n_meshes))
        basis = rot_y(rng.uniform(0, 360)) @ np.diag(rng.uniform(0.4, 1.6, size=3)) + rng.normal(0, 0.08, size=(3, 3))
        if rng.rand() < 0.3:
            basis[:, 0] = -basis[:, 0]                          # mirrored instance (negative determinant)
        ids = [int(rng.randint(0, n_mat)) for _ in meshes[m].surfaces]
        inst.append(Instance(m, transform12(basis, rng.uniform(-2.0, 2.0, size=3)), ids))
    tex = None
    if textured:
        tex = np.concatenate([checker_texture(32, 4), rng.randint(0, 256, size=(1, 32, 32, 4)).astype(np.uint8)])
    cam = CameraDesc(transform12(None, (0.0, 0.3, 6.5)), fov_deg=70.0)
    return Scene("fuzz%d" % seed, meshes, inst, np.stack(mats), cam, tex)


def checker_texture(res=64, cells=8) -> np.ndarray:
    y, x = np.mgrid[0:res, 0:res]
    c = (((x * cells // res) + (y * cells // res)) & 1).astype(np.uint8)
    t = np.zeros((1, res, res, 4), dtype=np.uint8)
    t[0, ..., 0] = 60 + 180 * c
    t[0, ..., 1] = 200 - 120 * c
    t[0, ..., 2] = 90 + 60 * c
    t[0, ..., 3] = 255
    return t


# --------------------------------------------------------------------------- camera block

def _t12_to_mat4(t12) -> np.ndarray:
    t = np.asarray(t12, dtype=np.float64)
    m = np.eye(4)
    m[:3, :3] = t[:9].reshape(3, 3)
    m[:3, 3] = t[9:12]
    return m


def perspective(fov_deg: float, aspect: float, near: float, far: float) -> np.ndarray:
    """Projection::create_perspective(fovy, aspect, near, far, false) (call site
    path_tracing_camera.cpp:134); godot-cpp is absent, this is its published algorithm."""
    rad = np.deg2rad(fov_deg / 2.0)
    cot = np.cos(rad) / np.sin(rad)
    dz = far - near
    p = np.zeros((4, 4))
    p[0, 0] = cot / aspect
    p[1, 1] = cot
    p[2, 2] = -(far + near) / dz
    p[3, 2] = -1.0
    p[2, 3] = -2.0 * near * far / dz
    return p


def camera_block(cam: CameraDesc, width: int, height: int, frame_index: int = 0) -> np.ndarray:
    """Camera::set_camera_transform (render_parameters.h:23-38): vp = P * M^-1, ivp = vp^-1,
    position = M.origin; matrices column-major (utils.h:39-49).  Computed in float64, stored f32 --
    this block is an opaque 160-byte input at the boundary."""
    m = _t12_to_mat4(cam.transform)
    p = perspective(cam.fov_deg, float(width) / float(height), cam.near, cam.far)
    vp = p @ np.linalg.inv(m)
    ivp = np.linalg.inv(vp)
    c = np.zeros((), dtype=wire.CAMERA)
    c["vp"] = vp.T.reshape(-1).astype(np.float32)    # column-major
    c["ivp"] = ivp.T.reshape(-1).astype(np.float32)
    c["position"] = (*m[:3, 3], 1.0)
    c["frame_index"] = frame_index
    c["near"] = cam.near
    c["far"] = cam.far
    return c


def view_projection(cam: CameraDesc, width: int, height: int) -> np.ndarray:
    """projection_matrix * Projection(get_global_transform().affine_inverse()) -- the `vp` of
    TemporalReprojection::render (temporal_reprojection.cpp:62; arguments from path_tracing_camera.cpp:220)."""
    return perspective(cam.fov_deg, float(width) / float(height), cam.near, cam.far) @ np.linalg.inv(_t12_to_mat4(cam.transform))


def temporal_delta(previous_vp: np.ndarray, vp: np.ndarray) -> np.ndarray:
    """temporal_reprojection.cpp:63,66: `Transform3D deltaMatrix = previous_vp * vp.inverse()` keeps the upper
    3x4 of the 4x4 product (Projection -> Transform3D drops the bottom row), and projection_to_float() of that
    Transform3D writes it back with bottom row 0 0 0 1.  Returns the 16 floats, column-major."""
    d = previous_vp @ np.linalg.inv(vp)
    d[3, :] = (0.0, 0.0, 0.0, 1.0)
    return d.T.reshape(-1).astype(np.float32)


def write_scene_file(scene: Scene, path: str) -> None:
    """Binary scene description read by tests/cpp/host_demo.cpp (the C++ host layer's test driver)."""
    import struct
    with open(path, "wb") as f:
        f.write(struct.pack("<II", 0x5354504A, len(scene.meshes)))
        for m in scene.meshes:
            f.write(struct.pack("<I", len(m.surfaces)))
            for s in m.surfaces:
                f.write(struct.pack("<II", len(s.vertices), len(s.indices)))
                f.write(s.vertices.tobytes()); f.write(s.normals.tobytes()); f.write(s.uvs.tobytes()); f.write(s.indices.tobytes())
        f.write(struct.pack("<I", len(scene.materials)))
        for m in scene.materials:
            f.write(struct.pack("<9fi", *m["albedo"][:3], m["metallic"], m["roughness"], *m["emission"][:3], m["emission"][3],
                                int(m["albedo_texture_index"])))
        f.write(struct.pack("<I", len(scene.instances)))
        for i in scene.instances:
            f.write(struct.pack("<I", i.mesh))
            f.write(np.asarray(i.transform, dtype=np.float32).tobytes())
            f.write(struct.pack("<I", len(i.material_ids)))
            f.write(np.asarray(i.material_ids, dtype=np.int32).tobytes())
        f.write(np.asarray(scene.camera.transform, dtype=np.float32).tobytes())
        f.write(struct.pack("<f", scene.camera.fov_deg))
