"""ctypes binding of the CPU parity oracle (``oracle/build/libpt_oracle.so``).

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, by ``__graft_entry__.smoke()``
and by the ``cpu_baseline`` leg of ``bench.py`` — never by the rendering
package.  The oracle accepts the same struct layouts as the GPU library, so the
ctypes classes of ``pathtracer_amd.abi`` are reused for the data contract only.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from pathtracer_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "build", "libpt_oracle.so")

RNG_COUNTER, RNG_MINSTD = 0, 1
MATH_LIBM, MATH_PORTABLE = 0, 1


class OrcStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("camera_samples", C.c_uint64), ("nodes_visited", C.c_uint64),
                ("triangles_tested", C.c_uint64), ("instances_tested", C.c_uint64),
                ("instances_entered", C.c_uint64), ("seconds", C.c_double), ("chains_cut", C.c_uint64)]


class OrcRay(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("direction", C.c_float * 3), ("payload", C.c_float * 3),
                ("depth", C.c_uint32), ("any_hit", C.c_uint32), ("key", C.c_uint32)]


def build(force: bool = False) -> str:
    """Compile the oracle with its Makefile if the library is missing or stale."""
    srcs = [os.path.join(_HERE, f) for f in ("pt_oracle.cc", "pt_oracle_core.h", "pt_oracle.h", "Makefile")]
    stale = force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = load_library(LIB_PATH)
    return _lib


def build_native(out_dir: str) -> str:
    """The same source compiled for the machine it runs on (-O3 -march=native, contraction still off): the build the
    CPU baseline of the benchmark is timed with.  Parity tests never use it — its arithmetic is the same, its code is
    tied to one CPU model.  Returns the library's path."""
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libpt_oracle_native.so")
    src = os.path.join(_HERE, "pt_oracle.cc")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["g++", "-O3", "-march=native", "-ffp-contract=off", "-fno-fast-math", "-std=c++17", "-fPIC", "-pthread",
                        "-shared", "-o", out, src], check=True, capture_output=True, timeout=300)
    return out


def load_library(path: str) -> C.CDLL:
    """Load one build of the oracle and declare its entry points."""
    L = C.CDLL(path)
    vp, P, f32p = C.c_void_p, C.POINTER, abi.c_float_p
    L.orc_scene_create.restype = vp
    L.orc_scene_create.argtypes = [P(abi.Model), P(abi.Envmap), P(abi.Settings)]
    L.orc_scene_destroy.argtypes = [vp]
    L.orc_set_settings.argtypes = [vp, P(abi.Settings)]
    L.orc_set_rng_mode.argtypes = [vp, C.c_int]
    L.orc_set_math_mode.argtypes = [vp, C.c_int]
    L.orc_update_camera_transform_state.argtypes = [vp, f32p, f32p, f32p, C.c_float, C.c_float, C.c_float, C.c_float]
    L.orc_set_camera_raw.argtypes = [vp, f32p, f32p]
    L.orc_get_camera_raw.argtypes = [vp, f32p, f32p]
    L.orc_get_scene_bound.argtypes = [vp, f32p, f32p]
    L.orc_render_frame.argtypes = [vp, C.c_int, C.c_int, f32p, C.c_int, P(OrcStats)]
    L.orc_render_frame_sums.argtypes = [vp, C.c_int, C.c_int, f32p, C.c_int, P(OrcStats)]
    L.orc_render_rows.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_int, P(OrcStats)]
    L.orc_render_hits.argtypes = [vp, C.c_int, C.c_int, C.c_uint32, P(abi.Hit), C.c_int]
    L.orc_intersect_rays.argtypes = [vp, C.c_uint64, f32p, f32p, abi.c_u8_p, P(abi.Hit)]
    L.orc_get_bvh_info.argtypes = [vp, C.c_uint32, P(abi.BvhInfo)]
    L.orc_subsample_positions.argtypes = [C.c_uint32, f32p]
    L.orc_kat_ray_triangle.argtypes = [f32p] * 5 + [C.c_int, f32p]
    L.orc_kat_bbox.argtypes = [f32p] * 5
    L.orc_kat_dop.argtypes = [f32p, C.c_int, f32p, f32p, f32p]
    L.orc_kat_sample_bilinear_rgba8.argtypes = [abi.c_u8_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, f32p]
    L.orc_kat_sample_environment.argtypes = [f32p, C.c_int, C.c_int, f32p, C.c_float, C.c_int, f32p]
    L.orc_kat_cos_dir.argtypes = [C.c_float, C.c_float, C.c_int, f32p]
    L.orc_kat_ggx.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int, f32p]
    L.orc_kat_schlick.argtypes = [f32p, f32p, C.c_float, f32p]
    L.orc_kat_v_smith.restype = C.c_float
    L.orc_kat_v_smith.argtypes = [C.c_float] * 3
    L.orc_kat_refract.argtypes = [f32p, f32p, C.c_float, f32p]
    L.orc_kat_fibonacci2d.argtypes = [C.c_int, C.c_float, f32p]
    L.orc_kat_srgb_to_linear.restype = C.c_float
    L.orc_kat_srgb_to_linear.argtypes = [C.c_float]
    L.orc_kat_linear_to_srgb.restype = C.c_float
    L.orc_kat_linear_to_srgb.argtypes = [C.c_float]
    L.orc_kat_float_pixel_to_srgb8.argtypes = [f32p, abi.c_u8_p]
    L.orc_kat_rng_camera_key.restype = C.c_uint32
    L.orc_kat_rng_camera_key.argtypes = [C.c_uint32] * 3
    L.orc_kat_rng_uniform.restype = C.c_float
    L.orc_kat_rng_uniform.argtypes = [C.c_uint32, C.c_uint32]
    L.orc_kat_rng_child.restype = C.c_uint32
    L.orc_kat_rng_child.argtypes = [C.c_uint32, C.c_uint32]
    L.orc_kat_on_hit.restype = C.c_int
    L.orc_kat_on_hit.argtypes = [vp, P(OrcRay), P(abi.Hit), f32p, P(OrcRay), C.c_int]
    return L


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(abi.c_float_p)


class OracleScene:
    """The oracle's counterpart of the reference ``Renderer`` with a scene loaded."""

    def __init__(self, model, env_rgba=None, settings: abi.Settings = None, library: C.CDLL = None):
        self._L = library if library is not None else lib()
        self._model = model
        cm = model.as_ctypes()
        env = None
        if env_rgba is not None:
            self._env = np.ascontiguousarray(env_rgba, np.float32)
            e = abi.Envmap(self._env.shape[1], self._env.shape[0], self._env.ctypes.data_as(abi.c_float_p))
            env = C.byref(e)
        self.settings = settings
        self._h = self._L.orc_scene_create(C.byref(cm), env, C.byref(settings))
        if not self._h:
            raise RuntimeError("orc_scene_create failed")

    def close(self):
        if self._h:
            self._L.orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_settings(self, settings: abi.Settings):
        self.settings = settings
        self._L.orc_set_settings(self._h, C.byref(settings))

    def set_rng_mode(self, mode: int):
        self._L.orc_set_rng_mode(self._h, mode)

    def set_math_mode(self, mode: int):
        self._L.orc_set_math_mode(self._h, mode)

    def update_camera_transform_state(self, position, direction, up, yfov, aspect, znear, zfar):
        _, p = _f32(position)
        _, d = _f32(direction)
        _, u = _f32(up)
        self._L.orc_update_camera_transform_state(self._h, p, d, u, yfov, aspect, znear, zfar)

    def set_camera_raw(self, ndc2world, origin):
        _, m = _f32(ndc2world)
        _, o = _f32(origin)
        self._L.orc_set_camera_raw(self._h, m, o)

    def get_camera_raw(self):
        m = np.zeros(16, np.float32)
        o = np.zeros(3, np.float32)
        self._L.orc_get_camera_raw(self._h, m.ctypes.data_as(abi.c_float_p), o.ctypes.data_as(abi.c_float_p))
        return m, o

    def get_scene_bound(self):
        a = np.zeros(3, np.float32)
        b = np.zeros(3, np.float32)
        self._L.orc_get_scene_bound(self._h, a.ctypes.data_as(abi.c_float_p), b.ctypes.data_as(abi.c_float_p))
        return a, b

    def render_frame(self, width, height, n_threads=0, sums=False, rows=None):
        out = np.zeros((height, width, 4), np.float32)
        st = OrcStats()
        p = out.ctypes.data_as(abi.c_float_p)
        if rows is not None:
            rc = self._L.orc_render_rows(self._h, width, height, rows[0], rows[1], p, n_threads, C.byref(st))
        elif sums:
            rc = self._L.orc_render_frame_sums(self._h, width, height, p, n_threads, C.byref(st))
        else:
            rc = self._L.orc_render_frame(self._h, width, height, p, n_threads, C.byref(st))
        if rc != 0:
            raise RuntimeError("oracle render failed")
        return out, st

    def render_hits(self, width, height, sample=0, n_threads=0):
        out = np.zeros((height, width), abi.HIT_DTYPE)
        rc = self._L.orc_render_hits(self._h, width, height, sample, out.ctypes.data_as(C.POINTER(abi.Hit)), n_threads)
        if rc != 0:
            raise RuntimeError("oracle render_hits failed")
        return out

    def intersect_rays(self, origins, directions, any_hit=None):
        o, po = _f32(origins)
        d, pd = _f32(directions)
        n = len(o)
        out = np.zeros(n, abi.HIT_DTYPE)
        pa = None
        if any_hit is not None:
            a = np.ascontiguousarray(any_hit, np.uint8)
            pa = a.ctypes.data_as(abi.c_u8_p)
        self._L.orc_intersect_rays(self._h, n, po, pd, pa, out.ctypes.data_as(C.POINTER(abi.Hit)))
        return out

    def bvh_info(self, mesh_index):
        info = abi.BvhInfo()
        if self._L.orc_get_bvh_info(self._h, mesh_index, C.byref(info)) != 0:
            raise RuntimeError("no BVH for mesh")
        return info

    def on_hit(self, ray: OrcRay, hit: abi.Hit, max_out=64):
        rad = np.zeros(3, np.float32)
        outs = (OrcRay * max_out)()
        n = self._L.orc_kat_on_hit(self._h, C.byref(ray), C.byref(hit), rad.ctypes.data_as(abi.c_float_p), outs, max_out)
        return rad, [outs[i] for i in range(min(n, max_out))]


def subsample_positions(spp: int) -> np.ndarray:
    out = np.zeros((spp, 2), np.float32)
    lib().orc_subsample_positions(spp, out.ctypes.data_as(abi.c_float_p))
    return out
