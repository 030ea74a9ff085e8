"""Host-side scene container: the reference's ``Model`` (``src/model_loader.h:59-65``) as numpy arrays.

``SceneModel`` owns the flat arrays the C ABI consumes (vertices, indices,
meshes, objects, materials, images, texels) plus the cameras a glTF file would
carry (``Camera``, ``model_loader.h:54-57``).  ``ModelBuilder`` assembles one
from meshes, materials and node transforms the way ``ModelLoader::constructModel``
(``src/model_loader.cc:55-214``) does for a glTF asset: one ``Mesh`` per
primitive, one ``Object`` per node, the default material appended last.
"""
from __future__ import annotations

import ctypes as C
import struct
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import abi


@dataclass
class Camera:
    """A perspective glTF camera after scene-graph flattening (position/direction/up in world space)."""
    position: Sequence[float]
    direction: Sequence[float]
    up: Sequence[float] = (0.0, 1.0, 0.0)
    yfov: float = np.radians(45.0)
    znear: float = 0.1
    zfar: float = 1000.0


def f0_dielectric(transmitted_ior: float, incident_ior: float = 1.0) -> np.float32:
    """``f0_dielectric`` of ``src/brdf.cc:70-72`` in binary32."""
    t, i = np.float32(transmitted_ior), np.float32(incident_ior)
    r = np.float32(np.float32(t - i) / np.float32(t + i))
    return np.float32(r * r)


@dataclass
class SceneModel:
    vertices: np.ndarray  # abi.VERTEX_DTYPE
    indices: np.ndarray   # uint32, mesh-relative
    meshes: np.ndarray    # abi.MESH_DTYPE
    objects: np.ndarray   # abi.OBJECT_DTYPE
    materials: np.ndarray  # abi.MATERIAL_DTYPE
    images: np.ndarray    # abi.IMAGE_DTYPE
    texels: np.ndarray    # uint8, (n_texels, 4)
    cameras: List[Camera] = field(default_factory=list)
    name: str = "scene"

    @property
    def n_triangles(self) -> int:
        return int(self.meshes["n_indices"].sum() // 3)

    def instanced_triangles(self) -> int:
        return int(sum(int(self.meshes["n_indices"][o]) // 3 for o in self.objects["mesh_index"]))

    def as_ctypes(self) -> abi.Model:
        """A ``ptgpu_model`` view over the arrays (valid while ``self`` is alive)."""
        m = abi.Model()
        m.vertices = self.vertices.ctypes.data_as(C.POINTER(abi.Vertex))
        m.n_vertices = len(self.vertices)
        m.indices = self.indices.ctypes.data_as(abi.c_u32_p)
        m.n_indices = len(self.indices)
        m.meshes = self.meshes.ctypes.data_as(C.POINTER(abi.Mesh))
        m.n_meshes = len(self.meshes)
        m.objects = self.objects.ctypes.data_as(C.POINTER(abi.Object))
        m.n_objects = len(self.objects)
        m.materials = self.materials.ctypes.data_as(C.POINTER(abi.Material))
        m.n_materials = len(self.materials)
        m.images = self.images.ctypes.data_as(C.POINTER(abi.Image))
        m.n_images = len(self.images)
        m.texels = self.texels.ctypes.data_as(abi.c_u8_p)
        m.n_texels = len(self.texels)
        return m

    # ---- flat binary container ("PTSC"), read by the C++ host's model loader -----------------
    MAGIC = b"PTSC0001"

    def save(self, path: str) -> None:
        with open(path, "wb") as f:
            f.write(self.MAGIC)
            f.write(struct.pack("<8Q", len(self.vertices), len(self.indices), len(self.meshes), len(self.objects),
                                len(self.materials), len(self.images), len(self.texels), len(self.cameras)))
            for arr in (self.vertices, self.indices, self.meshes, self.objects, self.materials, self.images,
                        self.texels):
                f.write(np.ascontiguousarray(arr).tobytes())
            for cam in self.cameras:
                f.write(struct.pack("<12f", *cam.position, *cam.direction, *cam.up, cam.yfov, cam.znear, cam.zfar))

    @classmethod
    def load(cls, path: str) -> "SceneModel":
        with open(path, "rb") as f:
            if f.read(8) != cls.MAGIC:
                raise ValueError(f"{path}: not a PTSC scene container")
            nv, ni, nm, no, nmat, nimg, ntex, ncam = struct.unpack("<8Q", f.read(64))

            def rd(dtype, n, shape=None):
                a = np.frombuffer(f.read(np.dtype(dtype).itemsize * n * (4 if shape else 1)), dtype=dtype).copy()
                return a.reshape(-1, 4) if shape else a
            vertices = rd(abi.VERTEX_DTYPE, nv)
            indices = rd(np.uint32, ni)
            meshes = rd(abi.MESH_DTYPE, nm)
            objects = rd(abi.OBJECT_DTYPE, no)
            materials = rd(abi.MATERIAL_DTYPE, nmat)
            images = rd(abi.IMAGE_DTYPE, nimg)
            texels = rd(np.uint8, ntex, shape=True)
            cams = []
            for _ in range(ncam):
                v = struct.unpack("<12f", f.read(48))
                cams.append(Camera(v[0:3], v[3:6], v[6:9], v[9], v[10], v[11]))
        return cls(vertices, indices, meshes, objects, materials, images, texels, cams, name=path)


def _normal_matrix(m: np.ndarray) -> np.ndarray:
    """``transpose(inverse(M))`` (``src/model_loader.cc:199``).  A matrix without an inverse (an object scaled to nothing)
    gives what glm's cofactor formula gives: every entry divided by a zero determinant, infinities and NaNs."""
    m = m.astype(np.float64)
    try:
        return np.linalg.inv(m).T
    except np.linalg.LinAlgError:
        with np.errstate(all="ignore"):
            cof = np.array([[(-1.0) ** (r + c) * np.linalg.det(np.delete(np.delete(m, r, 0), c, 1)) for c in range(4)] for r in range(4)])
            return cof / np.float64(0.0)


class ModelBuilder:
    """Assemble a :class:`SceneModel` from primitives, materials and node transforms."""

    def __init__(self, name: str = "scene"):
        self.name = name
        self._vertices: List[np.ndarray] = []
        self._indices: List[np.ndarray] = []
        self._meshes: List[tuple] = []
        self._objects: List[tuple] = []
        self._materials: List[dict] = []
        self._images: List[np.ndarray] = []
        self.cameras: List[Camera] = []
        self._nv = 0
        self._ni = 0

    # -- images / materials ---------------------------------------------------------------------
    def add_image(self, rgba8: np.ndarray) -> int:
        a = np.ascontiguousarray(rgba8, dtype=np.uint8)
        assert a.ndim == 3 and a.shape[2] == 4
        self._images.append(a)
        return len(self._images) - 1

    def add_material(self, base_color=(1, 1, 1, 1), emissive=(0, 0, 0), metallic=1.0, roughness=1.0,
                     base_color_texture=-1, metallic_roughness_texture=-1, normal_texture=-1, ior=1.5,
                     transmission=0.0, transmission_texture=-1, emissive_texture=-1, emissive_strength=1.0,
                     double_sided=False, attenuation_color=None, attenuation_distance=None,
                     alpha_mode="OPAQUE", alpha_cutoff=0.5) -> int:
        """Material mapping of ``src/model_loader.cc:95-123`` (Blender-style attenuation factor)."""
        has_volume = attenuation_color is not None
        att = np.zeros(3, np.float32)
        if has_volume:
            dist = np.float32(attenuation_distance if attenuation_distance is not None else np.inf)
            att = (np.float32(1.0) - np.asarray(attenuation_color, np.float32)) / np.maximum(np.float32(1e-5), dist)
        self._materials.append(dict(
            base_color_factor=base_color, emissive_factor=emissive, attenuation_factor=att,
            metallic_factor=metallic, roughness_factor=roughness, base_color_texture=base_color_texture,
            metallic_roughness_texture=metallic_roughness_texture, normal_texture=normal_texture, ior=ior,
            dielectric_f0=f0_dielectric(ior), transmission_factor=transmission,
            transmission_texture=transmission_texture, emissive_texture=emissive_texture,
            emissive_strength=emissive_strength, double_sided=int(double_sided), has_volume=int(has_volume),
            alpha_blending=int(alpha_mode == "BLEND"),
            alpha_cutoff=(alpha_cutoff if alpha_mode == "MASK" else -1.0)))
        return len(self._materials) - 1

    # -- geometry ---------------------------------------------------------------------------------
    def add_mesh(self, positions, normals, indices, uvs=None, tangents=None, material: Optional[int] = None) -> int:
        positions = np.asarray(positions, np.float32)
        normals = np.asarray(normals, np.float32)
        indices = np.asarray(indices, np.uint32).reshape(-1)
        assert len(indices) % 3 == 0 and (len(indices) == 0 or indices.max() < len(positions))
        n = len(positions)
        v = np.zeros(n, abi.VERTEX_DTYPE)
        v["position"] = positions
        v["normal"] = normals
        v["uv"] = 0.0 if uvs is None else np.asarray(uvs, np.float32)
        if tangents is None:
            # no-UV fallback of src/compute_tangents.cc:64-70: cross(normal, +-Y), w = 1
            diff = normals - np.array([0, 1, 0], np.float32)
            up = np.where((np.einsum("ij,ij->i", diff, diff) > 1e-4)[:, None], np.array([[0, 1, 0]], np.float32),
                          np.array([[0, -1, 0]], np.float32))
            t = np.cross(normals, up).astype(np.float32)
            tangents = np.concatenate([t, np.ones((n, 1), np.float32)], axis=1)
        v["tangent"] = np.asarray(tangents, np.float32)
        self._vertices.append(v)
        self._indices.append(indices)
        self._meshes.append((self._nv, n, self._ni, len(indices), -1 if material is None else material))
        self._nv += n
        self._ni += len(indices)
        return len(self._meshes) - 1

    def add_object(self, mesh: int, matrix=None) -> int:
        m = np.eye(4) if matrix is None else np.asarray(matrix, np.float64)
        self._objects.append((m, mesh))
        return len(self._objects) - 1

    def add_camera(self, cam: Camera) -> None:
        self.cameras.append(cam)

    def build(self) -> SceneModel:
        mats = list(self._materials)
        # default material appended at the last index (src/model_loader.cc:126)
        default_idx = len(mats)
        mb = ModelBuilder()
        mb.add_material()
        default = dict(mb._materials[0])
        default["dielectric_f0"] = np.float32(0.04)  # the literal of Material{} (model_loader.h:44), not f0_dielectric(1.5)
        mats.append(default)
        materials = np.zeros(len(mats), abi.MATERIAL_DTYPE)
        for i, m in enumerate(mats):
            for k, val in m.items():
                materials[k][i] = val
        meshes = np.zeros(len(self._meshes), abi.MESH_DTYPE)
        for i, (fv, nv, fi, ni, mat) in enumerate(self._meshes):
            meshes[i] = (fv, nv, fi, ni, default_idx if mat < 0 else mat)
        objects = np.zeros(len(self._objects), abi.OBJECT_DTYPE)
        for i, (m, mesh) in enumerate(self._objects):
            objects["model_matrix"][i] = m.T.reshape(-1).astype(np.float32)          # column-major
            objects["normal_matrix"][i] = _normal_matrix(m).T.reshape(-1).astype(np.float32)
            objects["mesh_index"][i] = mesh
        images = np.zeros(len(self._images), abi.IMAGE_DTYPE)
        off = 0
        tex = []
        for i, im in enumerate(self._images):
            images[i] = (im.shape[1], im.shape[0], off)
            off += im.shape[0] * im.shape[1]
            tex.append(im.reshape(-1, 4))
        texels = np.concatenate(tex) if tex else np.zeros((0, 4), np.uint8)
        vertices = np.concatenate(self._vertices) if self._vertices else np.zeros(0, abi.VERTEX_DTYPE)
        indices = np.concatenate(self._indices) if self._indices else np.zeros(0, np.uint32)
        return SceneModel(np.ascontiguousarray(vertices), np.ascontiguousarray(indices), meshes, objects, materials,
                          images, np.ascontiguousarray(texels), list(self.cameras), self.name)


# ---- small transform helpers (row-major 4x4 numpy, converted to column-major on build) ---------
def translate(x, y, z):
    m = np.eye(4)
    m[:3, 3] = (x, y, z)
    return m


def scale(x, y=None, z=None):
    y = x if y is None else y
    z = x if z is None else z
    return np.diag([x, y, z, 1.0])


def rotate(axis, degrees):
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    t = np.radians(degrees)
    c, s = np.cos(t), np.sin(t)
    x, y, z = a
    r = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s, 0],
                  [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s, 0],
                  [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c), 0],
                  [0, 0, 0, 1]])
    return r
