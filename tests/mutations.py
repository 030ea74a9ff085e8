"""Seeded corruptions of a GLB, a scene container, a PNG and a JPEG (byte flips, edits of the JSON chunk and of the
headers, truncations) plus hand-made files that lie about their sizes: input for the readers' robustness tests."""
from __future__ import annotations

import io
import json
import random
import struct

import numpy as np

from pathtracer_amd import gltf_export, scenes


def mutated_files(tmp_path, per_seed: int = 30):
    from PIL import Image as PIL
    rng = random.Random(7)
    glb = tmp_path / "s.glb"
    gltf_export.save_glb(scenes.tiny_scene(), str(glb))
    ptsc = tmp_path / "s.ptsc"
    scenes.tiny_scene().save(str(ptsc))
    img = (np.random.default_rng(0).uniform(0, 255, (23, 31, 3))).astype(np.uint8)
    seeds = {"glb": glb.read_bytes(), "ptsc": ptsc.read_bytes()}
    for fmt, kw in (("PNG", {}), ("JPEG", {"quality": 80, "subsampling": 2})):
        b = io.BytesIO()
        PIL.fromarray(img).save(b, fmt, **kw)
        seeds[fmt.lower()] = b.getvalue()
    paths = [str(glb), str(ptsc)]
    for name, data in seeds.items():
        json_len = struct.unpack_from("<I", data, 12)[0] if name == "glb" else 0
        for k in range(per_seed):
            m = bytearray(data)
            kind = rng.random()
            if kind < 0.4 and json_len:
                for _ in range(rng.randint(1, 4)):
                    m[rng.randrange(20, 20 + json_len)] = rng.choice(b'0123456789-[]{}",:ae ')
            elif kind < 0.8:
                for _ in range(rng.randint(1, 8)):
                    m[rng.randrange(len(m))] = rng.randrange(256)
            else:
                m = m[:rng.randrange(1, len(m))]
            p = tmp_path / f"{name}_{k}.{'glb' if name == 'glb' else name}"
            p.write_bytes(m)
            paths.append(str(p))
    # files whose numbers are chosen to wrap size computations
    jpeg = bytearray(seeds["jpeg"])
    at = jpeg.find(b"\xff\xc4")
    for tag, counts, n_values in (("a", bytes([255] * 16), 4080), ("b", bytes([3] + [0] * 14 + [253]), 256)):
        body = bytes([0x00]) + counts + bytes(n_values)
        p = tmp_path / f"dht_{tag}.jpeg"
        p.write_bytes(bytes(jpeg[:at]) + b"\xff\xc4" + struct.pack(">H", 2 + len(body)) + body + bytes(jpeg[at:]))
        paths.append(str(p))
    for field, value in ((0, 2 ** 61), (6, 2 ** 62), (1, 2 ** 40), (7, 2 ** 63)):
        bad = bytearray(seeds["ptsc"])
        struct.pack_into("<Q", bad, 8 + 8 * field, value)
        p = tmp_path / f"header_{field}.ptsc"
        p.write_bytes(bytes(bad))
        paths.append(str(p))
    json_len = struct.unpack_from("<I", seeds["glb"], 12)[0]
    doc = json.loads(seeds["glb"][20:20 + json_len])
    edits = [lambda d: d["bufferViews"][0].update(byteOffset=-8), lambda d: d["bufferViews"][0].update(byteStride=2 ** 40),
             lambda d: d["accessors"][0].update(count=2 ** 24 + 1), lambda d: d["accessors"][0].update(byteOffset=2 ** 62),
             lambda d: d["bufferViews"][0].update(byteLength=2.0 ** 63)]
    for k, edit in enumerate(edits):
        d = json.loads(json.dumps(doc))
        edit(d)
        text = json.dumps(d).encode()
        text += b" " * (-len(text) % 4)
        rest = seeds["glb"][20 + json_len:]
        out = b"glTF" + struct.pack("<II", 2, 12 + 8 + len(text) + len(rest)) + struct.pack("<I", len(text)) + b"JSON" + text + rest
        p = tmp_path / f"sizes_{k}.glb"
        p.write_bytes(out)
        paths.append(str(p))
    return paths
