"""The C++ glTF reader and image decoders of the host (the reference's ModelLoader, src/model_loader.cc:17-214, and
the stb decoding behind CPUTexture<sdr_pixel>, src/cpu_framebuffer.cc:36-83), driven through libptapp.so's C entry
points.  No GPU needed.  Pillow provides independent encoders/decoders to compare with."""
import ctypes as C
import io
import json
import os
import struct

import numpy as np
import pytest

from pathtracer_amd import gltf_export, scenes
from pathtracer_amd.model import SceneModel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pathtracer_amd", "lib", "libptapp.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB), reason="C++ host not built (run __graft_entry__.build())")
PIL = pytest.importorskip("PIL.Image")


@pytest.fixture(scope="module")
def lib():
    import pathtracer_amd._lib  # noqa: F401  (loads libptgpu.so, which libptapp.so depends on, from the same directory)
    L = C.CDLL(LIB)
    L.ptapp_convert_scene.restype = C.c_int
    L.ptapp_convert_scene.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
    L.ptapp_decode_image.restype = C.c_int
    L.ptapp_decode_image.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                     C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
    return L


def decode(lib, data: bytes):
    w, h, ch = C.c_int(), C.c_int(), C.c_int()
    err = C.create_string_buffer(256)
    if lib.ptapp_decode_image(data, len(data), w, h, ch, None, 0, err, 256):
        raise RuntimeError(err.value.decode())
    out = np.zeros((h.value, w.value, 4), np.uint8)
    assert lib.ptapp_decode_image(data, len(data), w, h, ch, out.ctypes.data, out.nbytes, err, 256) == 0
    return out, ch.value


def encoded(img, fmt, **kw):
    b = io.BytesIO()
    img.save(b, fmt, **kw)
    return b.getvalue()


def test_png_decoding_matches_pillow(lib):
    rng = np.random.default_rng(1)
    rgba = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    for mode, arr in (("RGBA", rgba), ("RGB", rgba[..., :3]), ("L", rgba[..., 0]), ("LA", rgba[..., [0, 3]])):
        img = PIL.fromarray(np.ascontiguousarray(arr), mode)
        got, ch = decode(lib, encoded(img, "PNG"))
        assert np.array_equal(got, np.asarray(img.convert("RGBA"))), mode
        assert ch == len(mode)
    pal = PIL.fromarray(rgba[..., :3], "RGB").quantize(colors=13)  # palette image with 4-bit indices
    got, _ = decode(lib, encoded(pal, "PNG", bits=4))
    assert np.array_equal(got, np.asarray(pal.convert("RGBA")))
    one = PIL.fromarray((rgba[..., 0] > 127).astype(np.uint8) * 255, "L").convert("1")  # 1-bit gray
    got, _ = decode(lib, encoded(one, "PNG"))
    assert np.array_equal(got, np.asarray(one.convert("RGBA")))
    g16 = rng.integers(0, 65536, (9, 11), dtype=np.uint16)  # 16-bit gray keeps its high byte
    got, _ = decode(lib, encoded(PIL.fromarray(g16), "PNG"))
    assert np.array_equal(got[..., 0], (g16 >> 8).astype(np.uint8)) and (got[..., 3] == 255).all()
    with pytest.raises(RuntimeError, match="neither PNG nor JPEG"):
        decode(lib, b"GIF89a" + bytes(32))
    with pytest.raises(RuntimeError, match="Unable to load image from memory"):
        decode(lib, encoded(PIL.fromarray(rgba, "RGBA"), "PNG")[:200])


def test_baseline_jpeg_decoding_close_to_pillow(lib):
    y, x = np.mgrid[0:96, 0:120]
    smooth = np.stack([127 + 100 * np.sin(x / 17.0), 127 + 90 * np.cos(y / 11.0), 60 + (x + y) * 0.8], -1)
    smooth = np.clip(smooth, 0, 255).astype(np.uint8)
    for sub, kw in ((0, {}), (2, {}), (1, {}), (2, {"restart_marker_blocks": 3})):
        data = encoded(PIL.fromarray(smooth, "RGB"), "JPEG", quality=92, subsampling=sub, **kw)
        got, ch = decode(lib, data)
        ref = np.asarray(PIL.open(io.BytesIO(data)).convert("RGBA")).astype(np.int32)
        d = np.abs(got.astype(np.int32) - ref)
        assert ch == 3 and d.max() <= 8 and d.mean() < 1.0, (sub, kw, d.max(), d.mean())
    gray = encoded(PIL.fromarray(smooth[..., 0], "L"), "JPEG", quality=90)
    got, ch = decode(lib, gray)
    ref = np.asarray(PIL.open(io.BytesIO(gray)).convert("RGBA")).astype(np.int32)
    assert ch == 1 and np.abs(got.astype(np.int32) - ref).max() <= 2


def test_progressive_jpeg_equals_its_sequential_twin(lib):
    """Progressive files (spectral selection + successive approximation, DC and AC refinement scans; what stb_image, the
    reference's decoder, reads too) carry the same quantised coefficients as a sequential file made from the same picture
    with the same tables, so both must decode to EXACTLY the same pixels here; and stay close to Pillow's decoder.
    Sizes that are not multiples of the MCU make the single-component scans cover fewer blocks than the interleaved
    ones; noise exercises long runs, refinement of many coefficients and end-of-band runs."""
    rng = np.random.default_rng(5)
    y, x = np.mgrid[0:61, 0:83]
    smooth = np.clip(np.stack([127 + 100 * np.sin(x / 9.0), 127 + 90 * np.cos(y / 7.0), 60 + (x + y) * 1.1], -1), 0, 255)
    noisy = np.clip(smooth + rng.normal(scale=40.0, size=smooth.shape), 0, 255).astype(np.uint8)
    for img, quality in ((smooth.astype(np.uint8), 90), (noisy, 95), (noisy, 35)):
        for sub in (0, 1, 2):
            for kw in ({}, {"restart_marker_blocks": 2}):
                seq = encoded(PIL.fromarray(img, "RGB"), "JPEG", quality=quality, subsampling=sub, **kw)
                prog = encoded(PIL.fromarray(img, "RGB"), "JPEG", quality=quality, subsampling=sub, progressive=True, **kw)
                assert b"\xff\xc2" in prog and b"\xff\xc2" not in seq
                a, ch = decode(lib, prog)
                b, _ = decode(lib, seq)
                assert ch == 3 and np.array_equal(a, b), (quality, sub, kw, np.abs(a.astype(int) - b).max())
                ref = np.asarray(PIL.open(io.BytesIO(prog)).convert("RGBA")).astype(np.int32)
                d = np.abs(a.astype(np.int32) - ref)
                assert d.mean() < 1.5, (quality, sub, d.max(), d.mean())
    gray = encoded(PIL.fromarray(noisy[..., 0], "L"), "JPEG", quality=85, progressive=True)
    got, ch = decode(lib, gray)
    ref = np.asarray(PIL.open(io.BytesIO(gray)).convert("RGBA")).astype(np.int32)
    assert ch == 1 and np.abs(got.astype(np.int32) - ref).max() <= 2


def convert(lib, src, dst):
    err = C.create_string_buffer(512)
    rc = lib.ptapp_convert_scene(str(src).encode(), str(dst).encode(), err, 512)
    if rc:
        raise RuntimeError(err.value.decode())
    return SceneModel.load(str(dst))


def test_glb_round_trip(lib, tmp_path):
    model = scenes.tiny_scene()
    gltf_export.save_glb(model, str(tmp_path / "tiny.glb"))
    back = convert(lib, tmp_path / "tiny.glb", tmp_path / "tiny.ptsc")
    assert np.array_equal(back.vertices, model.vertices) and np.array_equal(back.indices, model.indices)
    assert np.array_equal(back.meshes, model.meshes)
    assert np.array_equal(back.images, model.images) and np.array_equal(back.texels, model.texels)
    assert np.array_equal(back.objects["mesh_index"], model.objects["mesh_index"])
    assert np.array_equal(back.objects["model_matrix"], model.objects["model_matrix"])
    assert np.allclose(back.objects["normal_matrix"], model.objects["normal_matrix"], rtol=2e-5, atol=1e-6)
    assert len(back.materials) == len(model.materials)  # the default material is appended by the reader again
    for name in model.materials.dtype.names:
        a, b = back.materials[name], model.materials[name]
        if name == "attenuation_factor":
            assert np.allclose(a, b, rtol=1e-6, atol=1e-7)
        else:
            assert np.array_equal(a, b), name
    assert len(back.cameras) == len(model.cameras) == 1
    for f in ("position", "direction", "up"):
        assert np.allclose(getattr(back.cameras[0], f), getattr(model.cameras[0], f), atol=1e-6)
    assert back.cameras[0].yfov == pytest.approx(model.cameras[0].yfov, rel=1e-6)


def _minimal_gltf(tmp_path, prim_extra=None, node=None, material=None, mode=None):
    """A .gltf (JSON + external .bin) holding one triangle pair without indices, uv, tangents or material."""
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32)
    nrm = np.tile(np.array([[0, 0, 1]], np.float32), (6, 1))
    uv16 = (np.array([[0, 0], [1, 0], [0, 1], [1, 0], [1, 1], [0, 1]], np.float32) * 65535).astype(np.uint16)
    blob = pos.tobytes() + nrm.tobytes() + uv16.tobytes()
    (tmp_path / "geo.bin").write_bytes(blob)
    prim = {"attributes": {"POSITION": 0, "NORMAL": 1}}
    if prim_extra:
        prim["attributes"].update(prim_extra)
    if mode is not None:
        prim["mode"] = mode
    doc = {"asset": {"version": "2.0"}, "buffers": [{"uri": "geo.bin", "byteLength": len(blob)}],
           "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 72}, {"buffer": 0, "byteOffset": 72, "byteLength": 72},
                           {"buffer": 0, "byteOffset": 144, "byteLength": 24}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": 6, "type": "VEC3"},
                         {"bufferView": 1, "componentType": 5126, "count": 6, "type": "VEC3"},
                         {"bufferView": 2, "componentType": 5123, "normalized": True, "count": 6, "type": "VEC2"}],
           "meshes": [{"primitives": [prim, dict(prim)]}],
           "nodes": [{"children": [1], "translation": [1, 2, 3]}, node or {"mesh": 0, "scale": [2, 2, 2]}],
           "scenes": [{"nodes": [0]}]}
    if material:
        doc["materials"] = [material]
        doc["meshes"][0]["primitives"][0]["material"] = 0
    (tmp_path / "m.gltf").write_text(json.dumps(doc))
    return tmp_path / "m.gltf"


def test_gltf_defaults_and_hierarchy(lib, tmp_path):
    src = _minimal_gltf(tmp_path, material={"extensions": {"KHR_materials_volume": {"attenuationColor": [0.5, 1.0, 0.0]},
                                                           "KHR_materials_ior": {"ior": 1.33}}})
    m = convert(lib, src, tmp_path / "m.ptsc")
    assert len(m.meshes) == 2 and len(m.objects) == 2              # one Mesh per primitive, one Object per primitive of the node
    assert np.array_equal(m.indices, np.tile(np.arange(6, dtype=np.uint32), 2))   # GenerateMeshIndices
    assert len(m.materials) == 2 and m.meshes["material_index"].tolist() == [0, 1]  # second primitive -> default material
    assert (m.vertices["uv"] == 0).all()                           # no TEXCOORD_0
    # no uv: tangent = cross(normal, +Y), w = 1 (compute_tangents.cc:64-70)
    assert np.allclose(m.vertices["tangent"], [[-1, 0, 0, 1]])
    mat = m.materials[0]
    assert mat["has_volume"] == 1 and np.allclose(mat["attenuation_factor"], [0.0, 0.0, 0.0])  # infinite distance by default
    assert mat["ior"] == np.float32(1.33) and mat["dielectric_f0"] == pytest.approx(((1.33 - 1) / (1.33 + 1)) ** 2, rel=1e-6)
    assert mat["alpha_cutoff"] == -1.0 and mat["metallic_factor"] == 1.0 and mat["base_color_texture"] == -1
    dm = m.materials[1]                                             # Material{} defaults (model_loader.h:33-52)
    assert dm["ior"] == 1.5 and dm["dielectric_f0"] == np.float32(0.04) and dm["roughness_factor"] == 1.0
    world = m.objects["model_matrix"][0].reshape(4, 4).T            # parent translation x child scale
    assert np.allclose(world, [[2, 0, 0, 1], [0, 2, 0, 2], [0, 0, 2, 3], [0, 0, 0, 1]])
    nm = m.objects["normal_matrix"][0].reshape(4, 4).T
    assert np.allclose(nm, np.linalg.inv(world).T, atol=1e-6)
    # normalised 16-bit uv and generated tangents
    m = convert(lib, _minimal_gltf(tmp_path, prim_extra={"TEXCOORD_0": 2}), tmp_path / "m2.ptsc")
    assert np.allclose(m.vertices["uv"][:6], [[0, 0], [1, 0], [0, 1], [1, 0], [1, 1], [0, 1]])
    assert np.allclose(m.vertices["tangent"], [[1, 0, 0, 1]], atol=1e-6)


def test_gltf_errors(lib, tmp_path):
    with pytest.raises(RuntimeError, match="loadFromFile failed"):
        convert(lib, tmp_path / "missing.glb", tmp_path / "x.ptsc")
    with pytest.raises(RuntimeError, match="loadFromFile failed"):
        convert(lib, _minimal_gltf(tmp_path, mode=1), tmp_path / "x.ptsc")   # line primitives
    bad = tmp_path / "bad.glb"
    bad.write_bytes(b"glTF" + struct.pack("<II", 2, 1000) + b"\0" * 20)
    with pytest.raises(RuntimeError, match="loadFromFile failed"):
        convert(lib, bad, tmp_path / "x.ptsc")


def _patched_gltf(tmp_path, patch):
    src = _minimal_gltf(tmp_path)
    doc = json.loads(src.read_text())
    patch(doc)
    src.write_text(json.dumps(doc))
    return src


@pytest.mark.parametrize("patch", [
    lambda d: d["bufferViews"][0].update(byteOffset=-8),                       # negative offset
    lambda d: d["bufferViews"][0].update(byteOffset=1.5),                      # not an integer
    lambda d: d["bufferViews"][0].update(byteLength=2.0 ** 63),                # beyond what a double holds exactly
    lambda d: d["bufferViews"][0].update(byteStride=2 ** 40),                  # stride above the format's 252
    lambda d: d["accessors"][0].update(count=2 ** 24 + 1, byteOffset=0),       # with a wrapping product this passed
    lambda d: (d["bufferViews"][0].update(byteStride=4)),                      # elements would overlap
    lambda d: d["accessors"][0].update(byteOffset=2 ** 62),
    lambda d: d["accessors"][0].update(count=-3),
], ids=["neg-offset", "fractional", "huge-length", "huge-stride", "huge-count", "stride-below-element", "huge-accessor-offset",
        "neg-count"])
def test_gltf_rejects_sizes_that_would_wrap(lib, tmp_path, patch):
    with pytest.raises(RuntimeError, match="loadFromFile failed"):
        convert(lib, _patched_gltf(tmp_path, patch), tmp_path / "x.ptsc")


def test_scene_container_with_lying_header_is_rejected(lib, tmp_path):
    good = tmp_path / "good.ptsc"
    scenes.tiny_scene().save(str(good))
    raw = bytearray(good.read_bytes())
    for field, value in ((0, 2 ** 61), (6, 2 ** 62), (1, 2 ** 40), (7, 2 ** 63)):   # vertices, texels, indices, cameras
        bad = bytearray(raw)
        struct.pack_into("<Q", bad, 8 + 8 * field, value)
        (tmp_path / "bad.ptsc").write_bytes(bytes(bad))
        with pytest.raises(RuntimeError, match="loadFromFile failed"):
            convert(lib, tmp_path / "bad.ptsc", tmp_path / "x.ptsc")
    assert convert(lib, good, tmp_path / "again.ptsc") is not None


def test_jpeg_huffman_table_declaring_too_many_symbols(lib):
    """A DHT segment whose 16 counts sum to more than 256: rejected before any table is filled."""
    img = PIL.fromarray(np.random.default_rng(4).integers(0, 256, (16, 16, 3), dtype=np.uint8), "RGB")
    data = bytearray(encoded(img, "JPEG", quality=90))
    at = data.find(b"\xff\xc4")
    assert at > 0
    counts = bytes([255] * 16)                      # 4080 symbols declared
    body = bytes([0x00]) + counts + bytes(4080)
    seg = b"\xff\xc4" + struct.pack(">H", 2 + len(body)) + body
    with pytest.raises(RuntimeError, match="Huffman"):
        decode(lib, bytes(data[:at]) + seg + bytes(data[at:]))
    # counts that sum to 256 but put more codes on a length than it has
    counts = bytes([3] + [0] * 14 + [253])
    body = bytes([0x00]) + counts + bytes(256)
    seg = b"\xff\xc4" + struct.pack(">H", 2 + len(body)) + body
    with pytest.raises(RuntimeError, match="Huffman"):
        decode(lib, bytes(data[:at]) + seg + bytes(data[at:]))


# ---------------------------------------------------------------------------------------------------------------
# tangents for meshes with uv but without TANGENT (the MikkTSpace algorithm, src/compute_tangents.cc:9-62).
# Parity unpinned: the library is absent from the reference and its tests hold no tangent vectors; these are
# analytic cases and invariances of the published algorithm.
# ---------------------------------------------------------------------------------------------------------------
def tangents(lib, pos, nrm, uv, idx):
    pos = np.ascontiguousarray(pos, np.float32)
    nrm = np.ascontiguousarray(nrm, np.float32)
    uv = np.ascontiguousarray(uv, np.float32)
    idx = np.ascontiguousarray(idx, np.uint32).reshape(-1)
    out = np.zeros((len(pos), 4), np.float32)
    fp = C.POINTER(C.c_float)
    lib.ptapp_compute_tangents.restype = C.c_int
    lib.ptapp_compute_tangents.argtypes = [fp, fp, fp, C.c_size_t, C.POINTER(C.c_uint), C.c_size_t, fp]
    rc = lib.ptapp_compute_tangents(pos.ctypes.data_as(fp), nrm.ctypes.data_as(fp), uv.ctypes.data_as(fp), len(pos),
                                    idx.ctypes.data_as(C.POINTER(C.c_uint)), len(idx), out.ctypes.data_as(fp))
    assert rc == 0
    return out


def _grid(n, mirror=False):
    """(n+1)^2 vertices in the xz plane, normal +y, uv = (x, z) (or (-x, z)), triangles wound to face +y."""
    xs, zs = np.meshgrid(np.linspace(0, 1, n + 1), np.linspace(0, 1, n + 1), indexing="ij")
    pos = np.stack([xs.ravel(), np.zeros(xs.size), zs.ravel()], axis=1)
    nrm = np.tile([0.0, 1.0, 0.0], (len(pos), 1))
    uv = np.stack([-xs.ravel() if mirror else xs.ravel(), zs.ravel()], axis=1)
    tris = []
    for i in range(n):
        for j in range(n):
            a, b, c, d = i * (n + 1) + j, i * (n + 1) + j + 1, (i + 1) * (n + 1) + j, (i + 1) * (n + 1) + j + 1
            tris += [(a, b, c), (b, d, c)]  # (0,0,1) x (1,0,0) = +y
    return pos, nrm, uv, np.array(tris)


def test_tangents_of_a_plane_follow_u(lib):
    pos, nrm, uv, tris = _grid(4)
    t = tangents(lib, pos, nrm, uv, tris)
    assert np.allclose(t[:, :3], [1.0, 0.0, 0.0], atol=1e-6)
    # bitangent = cross(n, t) * w must point along +z, where v grows: cross(+y, +x) = -z, so w = -1
    assert (t[:, 3] == -1.0).all()
    pos, nrm, uv, tris = _grid(4, mirror=True)
    t = tangents(lib, pos, nrm, uv, tris)
    assert np.allclose(t[:, :3], [-1.0, 0.0, 0.0], atol=1e-6) and (t[:, 3] == 1.0).all()


def test_tangents_of_a_sphere_and_welding(lib):
    pos, nrm, tris, uv, analytic = scenes.uv_sphere(24, 48)
    t = tangents(lib, pos, nrm, uv, tris)
    assert np.isfinite(t).all()
    ring = (np.abs(pos[:, 1]) < 0.9 * np.abs(pos).max())  # away from the poles
    assert np.allclose(np.linalg.norm(t[ring, :3], axis=1), 1.0, atol=1e-5)
    assert np.abs((t[ring, :3] * nrm[ring]).sum(axis=1)).max() < 1e-5          # in the tangent plane
    # the direction in which u grows at constant v: around the y axis
    around = np.cross(np.tile([0.0, 1.0, 0.0], (len(pos), 1)), nrm)
    around /= np.maximum(np.linalg.norm(around, axis=1, keepdims=True), 1e-12)
    align = np.abs((t[ring, :3] * around[ring]).sum(axis=1))
    assert align.min() > 0.99
    assert len(np.unique(t[ring, 3])) == 1                                     # one orientation all over
    assert ((t[ring, :3] * analytic[ring, :3]).sum(axis=1) > 0.99).all()
    # the sign is the library's: +1 where the uv mapping keeps the orientation of the triangles' winding (here: outward
    # winding, positive uv area), whatever sign the scene generator chose for the tangents it ships itself
    assert (t[ring, 3] == 1.0).all()
    # the same surface as a triangle soup (every corner its own vertex): welding must give every copy the same tangent
    soup_idx = tris.reshape(-1)
    ts = tangents(lib, pos[soup_idx], nrm[soup_idx], uv[soup_idx], np.arange(len(soup_idx)))
    seam = np.isin(soup_idx, np.nonzero((uv[:, 0] == uv[:, 0].min()) | (uv[:, 0] == uv[:, 0].max()))[0])
    keep = ring[soup_idx] & ~seam
    assert np.allclose(ts[keep], t[soup_idx][keep], atol=1e-5)


def test_tangents_survive_degenerate_and_unmapped_triangles(lib):
    pos, nrm, uv, tris = _grid(3)
    base = tangents(lib, pos, nrm, uv, tris)
    extra = np.array([(0, 0, 5), (1, 1, 1)])                  # two corners at one position; all three at one
    t = tangents(lib, pos, nrm, uv, np.concatenate([tris, extra]))
    assert np.isfinite(t).all() and np.array_equal(t, base)
    uv0 = np.zeros_like(uv)                                   # no uv area anywhere: the library's default tangent space
    t0 = tangents(lib, pos, nrm, uv0, tris)
    assert np.isfinite(t0).all() and np.allclose(t0[:, :3], [1.0, 0.0, 0.0]) and (t0[:, 3] == -1.0).all()


def test_loaders_survive_mutated_files(tmp_path):
    """Corrupted input must end in an error return, never in a crash or a hang: 120 seeded mutations (byte flips in the
    JSON chunk, in the binary chunk and header, truncations) of a GLB, of PNG / JPEG streams and of a scene container,
    read in a child process."""
    import subprocess
    import sys
    from mutations import mutated_files
    paths = mutated_files(tmp_path)
    child = r"""
import ctypes as C, sys
sys.path.insert(0, sys.argv[1])
import pathtracer_amd._lib
L = C.CDLL(sys.argv[2])
L.ptapp_convert_scene.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
L.ptapp_decode_image.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p,
                                 C.c_size_t, C.c_char_p, C.c_size_t]
err = C.create_string_buffer(256)
for p in sys.argv[3:]:
    if p.endswith((".glb", ".ptsc")):
        L.ptapp_convert_scene(p.encode(), (p + ".out").encode(), err, 256)
    else:
        data = open(p, "rb").read()
        w, h, ch = C.c_int(), C.c_int(), C.c_int()
        if L.ptapp_decode_image(data, len(data), w, h, ch, None, 0, err, 256) == 0 and 0 < w.value * h.value < 1 << 24:
            buf = C.create_string_buffer(w.value * h.value * 4)
            L.ptapp_decode_image(data, len(data), w, h, ch, buf, len(buf), err, 256)
print("done")
"""
    r = subprocess.run([sys.executable, "-c", child, ROOT, LIB] + paths, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and r.stdout.strip().endswith("done"), (r.returncode, r.stderr[-400:])


def _interlaced_png(src: np.ndarray, ctype: int, depth: int = 8) -> bytes:
    """Adam7-interlaced PNG of an (h, w, channels) array (uint8, or uint16 for depth 16), filter type 0, built by
    hand: Pillow reads interlaced files but does not write them."""
    import zlib
    h, w = src.shape[:2]
    raw = b""
    for x0, y0, dx, dy in [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]:
        sub = src[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        for row in sub:
            raw += b"\x00" + (row.astype(">u2").tobytes() if depth == 16 else row.tobytes())

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1)) +
            chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


@pytest.mark.parametrize("ctype,channels", [(0, 1), (2, 3), (4, 2), (6, 4)])
@pytest.mark.parametrize("size", [(1, 1), (3, 2), (8, 8), (37, 23), (5, 40)])
def test_interlaced_png(lib, ctype, channels, size):
    """Adam7 interlacing (stb_image, which the reference decodes with, reads it): every colour type at 8 bits, sizes
    that leave some of the seven passes empty; Pillow must read the hand-built file as the same pixels."""
    w, h = size
    rng = np.random.default_rng(w * 100 + h + ctype)
    src = rng.integers(0, 256, (h, w, channels), dtype=np.uint8)
    png = _interlaced_png(src, ctype)
    expect = np.asarray(PIL.open(io.BytesIO(png)).convert("RGBA"))
    got, ch = decode(lib, png)
    assert ch == channels and np.array_equal(got, expect)


def test_interlaced_png_16_bit(lib):
    src = np.random.default_rng(3).integers(0, 65536, (9, 13, 3), dtype=np.uint16)
    got, ch = decode(lib, _interlaced_png(src, 2, depth=16))
    assert ch == 3 and np.array_equal(got[..., :3], (src >> 8).astype(np.uint8)) and (got[..., 3] == 255).all()
