"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, fails
loudly without a GPU, and the host-only helpers (sample table, image I/O, scene container) work."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import helpers as H
from pathtracer_amd import _lib, abi, image_io, scenes
from pathtracer_amd.model import SceneModel
from pathtracer_amd.renderer import RenderSettings, RayProgramMode, AccelerationStructureType

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "ptgpu.h")).read()
    declared = sorted(set(re.findall(r"\b(ptgpu_[a-z_]+)\s*\(", header)))
    assert declared == sorted(abi.PTGPU_SYMBOLS)
    lib = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_layouts_match_header():
    assert C.sizeof(abi.Vertex) == 48 and C.sizeof(abi.Mesh) == 20 and C.sizeof(abi.Object) == 132
    assert C.sizeof(abi.Material) == 100 and C.sizeof(abi.Image) == 16 and C.sizeof(abi.Hit) == 32
    assert C.sizeof(abi.Settings) == 52


@pytest.mark.skipif(not _no_gpu(), reason="checks the no-GPU failure path")
def test_create_fails_loudly_without_gpu():
    lib = _lib.load()
    h = C.c_void_p()
    rc = lib.ptgpu_create(0, C.byref(h))
    assert rc == -2 and not h
    assert b"HIP device" in lib.ptgpu_last_error(None) or lib.ptgpu_last_error(None)
    from pathtracer_amd.renderer import Renderer
    with pytest.raises(_lib.PtgpuError):
        Renderer(0)


def test_subsample_table_matches_golden():
    from pathtracer_amd.renderer import subsample_positions
    g = np.load(os.path.join(ROOT, "tests", "golden", "oracle_golden.npz"))
    for spp in (1, 64, 256, 300, 4096):
        assert np.array_equal(subsample_positions(spp), g[f"subsamples_{spp}"])


def test_settings_roundtrip():
    s = RenderSettings(300, 6, 0, 8, 0.5, RayProgramMode.PBR, AccelerationStructureType.BVH, 7, 1, 4)
    assert RenderSettings.from_abi(s.to_abi()) == s


def test_hdr_reader_and_writers(tmp_path):
    env = image_io.load_hdr(os.path.join(ROOT, "assets", "farmland_overcast_1k.hdr"))
    assert env.shape == (512, 1024, 4) and env.dtype == np.float32
    assert (env[..., 3] == 1).all() and np.isfinite(env).all() and env[..., :3].min() >= 0
    p = str(tmp_path / "x.hdr")
    image_io.write_hdr(p, env[:16, :32])
    back = image_io.load_hdr(p)
    assert np.array_equal(back, env[:16, :32])  # RGBE values re-encode exactly
    # PNG path: all four channels through the sRGB curve, alpha 1 -> 255 (cpu_framebuffer.h:52-59)
    from oracle import binding as ob
    px = np.array([[0.0, 0.002, 0.2, 1.0], [0.5, 1.5, -1.0, 1.0]], np.float32)
    mine = image_io.float_pixels_to_srgb8(px)
    for i in range(2):
        ref = np.zeros(4, np.uint8)
        ob.lib().orc_kat_float_pixel_to_srgb8(px[i].ctypes.data_as(abi.c_float_p), ref.ctypes.data_as(abi.c_u8_p))
        assert np.array_equal(mine[i], ref)
    image_io.save_framebuffer(np.tile(px[None], (3, 1, 1)), str(tmp_path / "x.png"))
    assert open(tmp_path / "x.png", "rb").read(8) == b"\x89PNG\r\n\x1a\n"


def test_scene_container_roundtrip(tmp_path):
    m = scenes.tiny_scene()
    p = str(tmp_path / "tiny.ptsc")
    m.save(p)
    b = SceneModel.load(p)
    for f in ("vertices", "indices", "meshes", "objects", "materials", "images", "texels"):
        assert np.array_equal(getattr(m, f), getattr(b, f)), f
    assert len(b.cameras) == 1 and np.allclose(b.cameras[0].position, m.cameras[0].position)
    # default material appended last (model_loader.cc:126)
    assert m.materials["metallic_factor"][-1] == 1.0 and m.materials["base_color_texture"][-1] == -1


def test_scenes_are_deterministic():
    a, b = scenes.readme_standin(0.05), scenes.readme_standin(0.05)
    assert np.array_equal(a.vertices, b.vertices) and np.array_equal(a.indices, b.indices)
    assert a.n_triangles > 1000 and len(a.objects) == 11
