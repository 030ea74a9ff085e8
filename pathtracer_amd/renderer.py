"""Python mirror of the reference's ``Renderer`` (``src/renderer.h:37-71``) on top of the C ABI.

Same method names and argument meaning as the C++ class: ``update_camera_transform_state``,
``load_scene``, ``render_frame``, ``set_render_settings``, ``get_render_settings``,
``get_scene_bound``.  ``render_frame`` raises ``RuntimeError`` when a component is missing, like
``renderer.cc:78-81``.  Everything executes in ``libptgpu.so`` on the GPU.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, replace
from enum import IntEnum
from typing import Optional

import numpy as np

from . import abi
from ._lib import PtgpuError, load
from .model import SceneModel


class RayProgramMode(IntEnum):  # src/arguments.h:10-16
    RayCaster = 0
    AmbientOcclusion = 1
    PBR = 2


class AccelerationStructureType(IntEnum):  # src/arguments.h:18-23
    Naive = 0
    BVH = 1
    DeviceLBVH = 2  # not in the reference: per-mesh trees built on the GPU (fast load, somewhat slower traversal)


@dataclass(frozen=True)
class RenderSettings:
    """``RenderSettings`` of ``src/render_settings.h:7-21`` (+ the sampler seed and the sample subset)."""
    samplesPerPixel: int = 1
    maxRayBounces: int = 4
    maxNewRaysPerBounce: int = 0
    maxTrianglesPerBVHLeaf: int = 8
    envmapRotation: float = 0.0  # radians
    programMode: RayProgramMode = RayProgramMode.PBR
    accelStructType: AccelerationStructureType = AccelerationStructureType.BVH
    seed: int = 0
    sampleBegin: int = 0
    sampleStep: int = 1
    singlePath: bool = False  # PBR keeps one of the rays a hit would spawn (same expectation, at most one ray per hit)
    wideNodes: bool = False   # read at load_scene: four-wide quantised nodes (same closest hits, about half the node visits)
    tileBegin: int = 0        # tile split: 8x8 tiles tileBegin, tileBegin + tileStep, ... are rendered, the others stay 0
    tileStep: int = 1

    def to_abi(self) -> abi.Settings:
        return abi.Settings(self.samplesPerPixel, self.maxRayBounces, self.maxNewRaysPerBounce,
                            self.maxTrianglesPerBVHLeaf, self.envmapRotation, int(self.programMode),
                            int(self.accelStructType), self.seed, self.sampleBegin, self.sampleStep,
                            (abi.FLAG_SINGLE_PATH if self.singlePath else 0) | (abi.FLAG_WIDE_NODES if self.wideNodes else 0),
                            self.tileBegin, self.tileStep)

    @staticmethod
    def from_abi(s: abi.Settings) -> "RenderSettings":
        return RenderSettings(s.samples_per_pixel, s.max_ray_bounces, s.max_new_rays_per_bounce,
                              s.max_triangles_per_bvh_leaf, s.envmap_rotation, RayProgramMode(s.program_mode),
                              AccelerationStructureType(s.accel_struct_type), s.seed, s.sample_begin, s.sample_step,
                              bool(s.flags & abi.FLAG_SINGLE_PATH), bool(s.flags & abi.FLAG_WIDE_NODES), s.tile_begin, s.tile_step or 1)


@dataclass
class Perspective:
    """Stand-in for ``fastgltf::Camera::Perspective`` (yfov, optional aspectRatio, znear, optional zfar)."""
    yfov: float
    znear: float = 0.1
    aspectRatio: Optional[float] = None
    zfar: Optional[float] = None


class CPUFrameBuffer:
    """``CPUFrameBuffer`` (``src/cpu_framebuffer.h:135-155``): width x height RGBA float pixels."""

    def __init__(self, width: int = 1, height: int = 1):
        self.data = np.zeros((height, width, 4), np.float32)

    def width(self) -> int:
        return self.data.shape[1]

    def height(self) -> int:
        return self.data.shape[0]

    def clear(self, color=(0.0, 0.0, 0.0, 1.0)) -> None:
        self.data[...] = np.asarray(color, np.float32)

    def at(self, x: int, y: int) -> np.ndarray:
        return self.data[y, x]

    def save_to_file(self, path: str) -> None:
        from .image_io import save_framebuffer
        save_framebuffer(self.data, path)


def _fp(a):
    return a.ctypes.data_as(abi.c_float_p)


class Renderer:
    def __init__(self, device: int = 0):
        self._L = load()
        h = C.c_void_p()
        rc = self._L.ptgpu_create(device, C.byref(h))
        if rc != 0:
            raise PtgpuError(rc, self._L.ptgpu_last_error(None).decode())
        self._h = h
        self.device = device
        self._model = None
        self._settings = RenderSettings()
        self.last_stats: Optional[abi.Stats] = None

    # -- plumbing -------------------------------------------------------------------------------
    def _check(self, rc: int) -> None:
        if rc != 0:
            raise PtgpuError(rc, self._L.ptgpu_last_error(self._h).decode())

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.ptgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the reference interface ------------------------------------------------------------------
    def set_render_settings(self, settings: RenderSettings) -> None:
        s = settings.to_abi()
        self._check(self._L.ptgpu_set_render_settings(self._h, C.byref(s)))
        self._settings = settings

    def get_render_settings(self) -> RenderSettings:
        s = abi.Settings()
        self._check(self._L.ptgpu_get_render_settings(self._h, C.byref(s)))
        return RenderSettings.from_abi(s)

    def load_scene(self, model: SceneModel, envmap: Optional[np.ndarray] = None) -> None:
        """``envmap``: (H, W, 4) float32 rows, or None for the 1x1 white default of ``main.cc:34-37``."""
        cm = model.as_ctypes()
        env_ref = None
        if envmap is not None:
            env_arr = np.ascontiguousarray(envmap, np.float32)
            assert env_arr.ndim == 3 and env_arr.shape[2] == 4
            e = abi.Envmap(env_arr.shape[1], env_arr.shape[0], _fp(env_arr))
            env_ref = C.byref(e)
        self._check(self._L.ptgpu_load_scene(self._h, C.byref(cm), env_ref))
        self._model = model

    def load_scene_from(self, source: "Renderer") -> None:
        """``ptgpu_load_scene_from``: take over the scene another renderer has loaded, device to device — the acceleration
        structure is built once however many GPUs render it."""
        self._check(self._L.ptgpu_load_scene_from(self._h, source._h))
        self._model = source._model
        self._settings = self.get_render_settings()

    def set_options(self, **fields) -> None:
        """``ptgpu_set_options``: fields of ``abi.Options`` by name; the others keep their current value."""
        o = abi.Options()
        self._check(self._L.ptgpu_get_options(self._h, C.byref(o)))
        for k, v in fields.items():
            if not hasattr(o, k):
                raise AttributeError(k)
            setattr(o, k, v)
        self._check(self._L.ptgpu_set_options(self._h, C.byref(o)))

    def get_options(self) -> abi.Options:
        o = abi.Options()
        self._check(self._L.ptgpu_get_options(self._h, C.byref(o)))
        return o

    def update_camera_transform_state(self, position, direction, up, perspective: Perspective) -> None:
        p = np.ascontiguousarray(position, np.float32)
        d = np.ascontiguousarray(direction, np.float32)
        u = np.ascontiguousarray(up, np.float32)
        aspect = perspective.aspectRatio if perspective.aspectRatio is not None else -1.0
        zfar = perspective.zfar if perspective.zfar is not None else -1.0
        self._check(self._L.ptgpu_update_camera_transform_state(self._h, _fp(p), _fp(d), _fp(u), perspective.yfov,
                                                                aspect, perspective.znear, zfar))

    def render_frame(self, framebuffer: CPUFrameBuffer) -> None:
        st = abi.Stats()
        out = framebuffer.data
        try:
            self._check(self._L.ptgpu_render_frame(self._h, framebuffer.width(), framebuffer.height(), _fp(out),
                                                   C.byref(st)))
        except PtgpuError as e:
            if e.code == -4:  # renderer.cc:79-81
                raise RuntimeError("One of components is nullptr in Renderer::render_frame()") from e
            raise
        self.last_stats = st

    def get_scene_bound(self):
        a = np.zeros(3, np.float32)
        b = np.zeros(3, np.float32)
        self._check(self._L.ptgpu_get_scene_bound(self._h, _fp(a), _fp(b)))
        return a, b

    # -- additions of the GPU path ----------------------------------------------------------------
    def set_camera_raw(self, ndc2world, origin) -> None:
        m = np.ascontiguousarray(ndc2world, np.float32).reshape(16)
        o = np.ascontiguousarray(origin, np.float32)
        self._check(self._L.ptgpu_set_camera_raw(self._h, _fp(m), _fp(o)))

    def get_camera_raw(self):
        m = np.zeros(16, np.float32)
        o = np.zeros(3, np.float32)
        self._check(self._L.ptgpu_get_camera_raw(self._h, _fp(m), _fp(o)))
        return m, o

    def render_counted(self, width: int, height: int):
        out = np.zeros((height, width, 4), np.float32)
        st = abi.Stats()
        self._check(self._L.ptgpu_render_counted(self._h, width, height, _fp(out), C.byref(st)))
        self.last_stats = st
        return out, st

    def render_frame_sums(self, width: int, height: int, device_ptr: int, stream: int = 0) -> abi.Stats:
        """Leave per-pixel sums of this renderer's sample subset in device memory (multi-GPU sample split)."""
        st = abi.Stats()
        self._check(self._L.ptgpu_render_frame_sums(self._h, width, height, C.c_void_p(device_ptr),
                                                    C.c_void_p(stream) if stream else None, C.byref(st)))
        self.last_stats = st
        return st

    def resolve_sums(self, width: int, height: int, device_ptr: int, total_samples: int, stream: int = 0) -> None:
        self._check(self._L.ptgpu_resolve_sums(self._h, width, height, C.c_void_p(device_ptr), total_samples,
                                               C.c_void_p(stream) if stream else None))

    def render_hits(self, width: int, height: int, sample: int = 0) -> np.ndarray:
        out = np.zeros((height, width), abi.HIT_DTYPE)
        self._check(self._L.ptgpu_render_hits(self._h, width, height, sample, out.ctypes.data_as(C.POINTER(abi.Hit))))
        return out

    def intersect_rays(self, origins, directions, any_hit=None) -> np.ndarray:
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        assert o.shape == d.shape
        out = np.zeros(len(o), abi.HIT_DTYPE)
        pa = None
        if any_hit is not None:
            a = np.ascontiguousarray(any_hit, np.uint8)
            assert len(a) == len(o)
            pa = a.ctypes.data_as(abi.c_u8_p)
        self._check(self._L.ptgpu_intersect_rays(self._h, len(o), _fp(o), _fp(d), pa,
                                                 out.ctypes.data_as(C.POINTER(abi.Hit))))
        return out

    def intersect_rays_static(self, n: int, d_origins: int, d_directions: int, d_any_hit: int, d_out: int, stream: int = 0,
                              wait: bool = True) -> None:
        """``ptgpu_intersect_rays_static``: the same batch on device pointers (integers), without any allocation inside
        the library; at most ``abi.STATIC_MAX_RAYS`` rays per call."""
        self._check(self._L.ptgpu_intersect_rays_static(self._h, n, d_origins or None, d_directions or None, d_any_hit or None,
                                                        d_out or None, stream or None, 1 if wait else 0))

    def bvh_info(self, mesh_index: int) -> abi.BvhInfo:
        info = abi.BvhInfo()
        self._check(self._L.ptgpu_get_bvh_info(self._h, mesh_index, C.byref(info)))
        return info


def subsample_positions(spp: int) -> np.ndarray:
    out = np.zeros((spp, 2), np.float32)
    rc = load().ptgpu_subsample_positions(spp, _fp(out))
    if rc != 0:
        raise PtgpuError(rc, "ptgpu_subsample_positions")
    return out
