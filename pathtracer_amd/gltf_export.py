"""Write a :class:`SceneModel` as a glTF 2.0 binary (``.glb``).

The inverse of what ``ModelLoader`` reads (``src/model_loader.cc:55-214``): one glTF mesh with one primitive per
``Mesh``, one node with a ``matrix`` per ``Object``, materials with the extensions the reference accepts
(``KHR_materials_ior/_transmission/_volume/_emissive_strength``), images embedded as PNG, cameras as nodes whose
matrix carries position / direction / up the way ``Viewer::snap_to_camera`` reads them (``src/viewer.cc:61-63``).
Used by the tests of the C++ glTF reader and to hand scenes to other glTF consumers.
"""
from __future__ import annotations

import io
import json
import struct
import zlib

import numpy as np

from .model import SceneModel


def _png_bytes(rgba8: np.ndarray) -> bytes:
    h, w = rgba8.shape[:2]
    raw = np.zeros((h, 1 + w * 4), np.uint8)
    raw[:, 1:] = np.ascontiguousarray(rgba8, np.uint8).reshape(h, w * 4)

    def chunk(tag: bytes, payload: bytes) -> bytes:
        return struct.pack(">I", len(payload)) + tag + payload + struct.pack(">I", zlib.crc32(tag + payload) & 0xffffffff)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0))
            + chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)) + chunk(b"IEND", b""))


def _f(x) -> float:
    """A binary32 value as the JSON number that parses back to exactly that value."""
    return float(np.float32(x))


def save_glb(model: SceneModel, path: str, image_encoder=None) -> None:
    """``image_encoder(rgba8) -> (bytes, mime)`` overrides the embedded PNG encoding (e.g. to embed JPEG)."""
    bin_parts = []
    views, accessors = [], []
    offset = 0

    def add_view(data: bytes, target=None) -> int:
        nonlocal offset
        pad = (-offset) % 4
        if pad:
            bin_parts.append(b"\0" * pad)
            offset += pad
        v = {"buffer": 0, "byteOffset": offset, "byteLength": len(data)}
        if target:
            v["target"] = target
        views.append(v)
        bin_parts.append(data)
        offset += len(data)
        return len(views) - 1

    def add_accessor(arr: np.ndarray, ctype: int, atype: str, target=None, minmax=False) -> int:
        a = {"bufferView": add_view(np.ascontiguousarray(arr).tobytes(), target), "componentType": ctype,
             "count": int(arr.shape[0]), "type": atype}
        if minmax:
            a["min"] = [float(v) for v in arr.min(axis=0)]
            a["max"] = [float(v) for v in arr.max(axis=0)]
        accessors.append(a)
        return len(accessors) - 1

    images, textures = [], []
    for im in model.images:
        w, h, off = int(im["width"]), int(im["height"]), int(im["texel_offset"])
        rgba = model.texels[off:off + w * h].reshape(h, w, 4)
        data, mime = image_encoder(rgba) if image_encoder else (_png_bytes(rgba), "image/png")
        images.append({"bufferView": add_view(data), "mimeType": mime})
        textures.append({"source": len(images) - 1})

    n_default = len(model.materials) - 1  # the loader appends its default material last: not exported
    materials = []
    for m in model.materials[:n_default]:
        mat = {"pbrMetallicRoughness": {"baseColorFactor": [_f(v) for v in m["base_color_factor"]],
                                        "metallicFactor": _f(m["metallic_factor"]),
                                        "roughnessFactor": _f(m["roughness_factor"])},
               "emissiveFactor": [_f(v) for v in m["emissive_factor"]],
               "doubleSided": bool(m["double_sided"]), "extensions": {}}
        if m["base_color_texture"] >= 0:
            mat["pbrMetallicRoughness"]["baseColorTexture"] = {"index": int(m["base_color_texture"])}
        if m["metallic_roughness_texture"] >= 0:
            mat["pbrMetallicRoughness"]["metallicRoughnessTexture"] = {"index": int(m["metallic_roughness_texture"])}
        if m["normal_texture"] >= 0:
            mat["normalTexture"] = {"index": int(m["normal_texture"])}
        if m["emissive_texture"] >= 0:
            mat["emissiveTexture"] = {"index": int(m["emissive_texture"])}
        if m["alpha_blending"]:
            mat["alphaMode"] = "BLEND"
        elif m["alpha_cutoff"] >= 0:
            mat["alphaMode"] = "MASK"
            mat["alphaCutoff"] = _f(m["alpha_cutoff"])
        ext = mat["extensions"]
        ext["KHR_materials_ior"] = {"ior": _f(m["ior"])}
        if m["emissive_strength"] != 1.0:
            ext["KHR_materials_emissive_strength"] = {"emissiveStrength": _f(m["emissive_strength"])}
        if m["transmission_factor"] != 0.0 or m["transmission_texture"] >= 0:
            t = {"transmissionFactor": _f(m["transmission_factor"])}
            if m["transmission_texture"] >= 0:
                t["transmissionTexture"] = {"index": int(m["transmission_texture"])}
            ext["KHR_materials_transmission"] = t
        if m["has_volume"]:
            # attenuation = (1 - colour) / distance in the reader: distance 1 keeps the factor as it is
            ext["KHR_materials_volume"] = {"attenuationColor": [_f(np.float32(1.0) - np.float32(v)) for v in m["attenuation_factor"]],
                                           "attenuationDistance": 1.0}
        materials.append(mat)

    meshes = []
    for me in model.meshes:
        v0, nv, i0, ni = int(me["first_vertex"]), int(me["n_vertices"]), int(me["first_index"]), int(me["n_indices"])
        vtx = model.vertices[v0:v0 + nv]
        attrs = {"POSITION": add_accessor(vtx["position"], 5126, "VEC3", 34962, minmax=True),
                 "NORMAL": add_accessor(vtx["normal"], 5126, "VEC3", 34962),
                 "TANGENT": add_accessor(vtx["tangent"], 5126, "VEC4", 34962),
                 "TEXCOORD_0": add_accessor(vtx["uv"], 5126, "VEC2", 34962)}
        prim = {"attributes": attrs, "indices": add_accessor(model.indices[i0:i0 + ni].astype(np.uint32), 5125, "SCALAR", 34963)}
        if int(me["material_index"]) != n_default:
            prim["material"] = int(me["material_index"])
        meshes.append({"primitives": [prim]})

    nodes = []
    for o in model.objects:
        nodes.append({"mesh": int(o["mesh_index"]), "matrix": [_f(v) for v in o["model_matrix"]]})
    cameras = []
    for c in model.cameras:
        pos, d, up = (np.asarray(v, np.float32) for v in (c.position, c.direction, c.up))
        right = np.cross(d, up).astype(np.float32)
        mat = np.zeros((4, 4), np.float32)  # columns: right, up, -direction, position
        mat[0, :3], mat[1, :3], mat[2, :3], mat[3, :3] = right, up, -d, pos
        mat[3, 3] = 1.0
        cameras.append({"type": "perspective", "perspective": {"yfov": _f(c.yfov), "znear": _f(c.znear), "zfar": _f(c.zfar)}})
        nodes.append({"camera": len(cameras) - 1, "matrix": [_f(v) for v in mat.reshape(-1)]})

    doc = {"asset": {"version": "2.0", "generator": "pathtracer_amd"}, "scene": 0,
           "scenes": [{"nodes": list(range(len(nodes)))}], "nodes": nodes, "meshes": meshes, "materials": materials,
           "accessors": accessors, "bufferViews": views,
           "extensionsUsed": ["KHR_materials_ior", "KHR_materials_transmission", "KHR_materials_volume",
                              "KHR_materials_emissive_strength"]}
    if images:
        doc["images"], doc["textures"] = images, textures
    if cameras:
        doc["cameras"] = cameras
    bin_blob = b"".join(bin_parts)
    bin_blob += b"\0" * ((-len(bin_blob)) % 4)
    doc["buffers"] = [{"byteLength": len(bin_blob)}]
    js = json.dumps(doc, separators=(",", ":")).encode()
    js += b" " * ((-len(js)) % 4)
    with open(path, "wb") as f:
        f.write(struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(bin_blob)))
        f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
        f.write(struct.pack("<II", len(bin_blob), 0x004E4942) + bin_blob)
