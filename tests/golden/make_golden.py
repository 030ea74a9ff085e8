"""Generate tests/golden/oracle_golden.npz from the CPU oracle (oracle/pt_oracle.cc).

These vectors pin the ORACLE (and, through the GPU parity tests, the HIP path) to the values it
produced when it was written; they do not come from the reference, which holds no vectors for this
path and cannot be built here ("parity unpinned", see oracle/pt_oracle.h).  Run from the repo root:
    python tests/golden/make_golden.py
"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers as H  # noqa: E402
from oracle import binding as ob  # noqa: E402
from pathtracer_amd import abi, scenes  # noqa: E402


def digest(arr) -> str:
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


def f32p(a):
    return a.ctypes.data_as(abi.c_float_p)


def kat_tables(seed=99):
    L = ob.lib()
    rng = np.random.default_rng(seed)
    out = {}
    # ray / triangle: random cases plus edge-on, vertex-on, parallel and behind-origin cases
    n = 256
    o = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    tri = rng.uniform(-1, 1, (n, 3, 3)).astype(np.float32)
    target = (tri * rng.dirichlet((1, 1, 1), n).astype(np.float32)[:, :, None]).sum(1)
    d = target - o
    d[n // 2:] = rng.normal(size=(n - n // 2, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    o[0], d[0] = (0.25, 0.25, 1), (0, 0, -1)
    tri[0] = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]
    o[1], d[1], tri[1] = (0.5, 0.0, 1), (0, 0, -1), tri[0]          # on an edge
    o[2], d[2], tri[2] = (0.0, 0.0, 1), (0, 0, -1), tri[0]          # on a vertex
    o[3], d[3], tri[3] = (0.25, 0.25, 1), (1, 0, 0), tri[0]         # parallel (det == 0)
    o[4], d[4], tri[4] = (0.25, 0.25, -1), (0, 0, -1), tri[0]       # behind the origin (t < 0)
    o[5], d[5], tri[5] = (0.25, 0.25, -1), (0, 0, 1), tri[0]        # back face
    res = np.zeros((n, 2, 5), np.float32)
    for i in range(n):
        for cull in (0, 1):
            L.orc_kat_ray_triangle(f32p(o[i]), f32p(d[i]), f32p(tri[i, 0]), f32p(tri[i, 1]), f32p(tri[i, 2]), cull,
                                   f32p(res[i, cull]))
    out.update(tri_o=o, tri_d=d, tri_p=tri, tri_res=res)
    # boxes and DOPs
    n = 128
    lo = rng.uniform(-1, 0, (n, 3)).astype(np.float32)
    hi = (lo + rng.uniform(0, 2, (n, 3))).astype(np.float32)
    hi[:8, 1] = lo[:8, 1]  # flat boxes
    bo = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    bd = rng.normal(size=(n, 3))
    bd[8:16, 0] = 0.0  # axis-parallel rays (infinite inverse direction)
    bd = (bd / np.linalg.norm(bd, axis=1, keepdims=True)).astype(np.float32)
    bres = np.zeros((n, 4), np.float32)
    dres = np.zeros((n, 4), np.float32)
    pts = rng.uniform(-1, 1, (n, 6, 3)).astype(np.float32)
    for i in range(n):
        L.orc_kat_bbox(f32p(lo[i]), f32p(hi[i]), f32p(bo[i]), f32p(bd[i]), f32p(bres[i]))
        L.orc_kat_dop(f32p(pts[i]), 6, f32p(bo[i]), f32p(bd[i]), f32p(dres[i]))
    out.update(box_lo=lo, box_hi=hi, box_o=bo, box_d=bd, box_res=bres, dop_pts=pts, dop_res=dres)
    # textures
    tex = rng.integers(0, 256, (5, 7, 4), dtype=np.uint8)
    uv = rng.uniform(-2.5, 2.5, (64, 2)).astype(np.float32)
    uv[:4] = [[0, 0], [1, 1], [0.5 / 7, 0.5 / 5], [-0.0001, 0.9999]]
    tres = np.zeros((64, 2, 4), np.float32)
    for i in range(64):
        for srgb in (0, 1):
            L.orc_kat_sample_bilinear_rgba8(tex.ctypes.data_as(abi.c_u8_p), 7, 5, float(uv[i, 0]), float(uv[i, 1]), srgb,
                                            f32p(tres[i, srgb]))
    env = scenes.synthetic_env(16, 8)
    ed = rng.normal(size=(64, 3))
    ed[:3] = [[0, 1, 0], [0, -1, 0], [-1, 0, 0]]
    ed = (ed / np.linalg.norm(ed, axis=1, keepdims=True)).astype(np.float32)
    eres = np.zeros((64, 2, 4), np.float32)
    for i in range(64):
        for mm in (0, 1):
            L.orc_kat_sample_environment(f32p(env), 16, 8, f32p(ed[i]), 0.3, mm, f32p(eres[i, mm]))
    out.update(tex=tex, tex_uv=uv, tex_res=tres, env=env, env_dir=ed, env_res=eres)
    # BRDF helpers
    xi = rng.uniform(0, 1, (64, 2)).astype(np.float32)
    xi[:3] = [[0, 0], [0.999999, 0.5], [0.5, 0.999999]]
    cosd = np.zeros((64, 2, 3), np.float32)
    ggx = np.zeros((64, 2, 3), np.float32)
    a = rng.uniform(0.001, 1, 64).astype(np.float32)
    for i in range(64):
        for mm in (0, 1):
            L.orc_kat_cos_dir(float(xi[i, 0]), float(xi[i, 1]), mm, f32p(cosd[i, mm]))
            L.orc_kat_ggx(float(xi[i, 0]), float(xi[i, 1]), float(a[i]), mm, f32p(ggx[i, mm]))
    f0 = rng.uniform(0, 1, (64, 3)).astype(np.float32)
    f90 = np.ones((64, 3), np.float32)
    c = rng.uniform(0, 1, 64).astype(np.float32)
    sch = np.zeros((64, 3), np.float32)
    vs = np.zeros(64, np.float32)
    refr = np.zeros((64, 3), np.float32)
    I = rng.normal(size=(64, 3))
    I = (I / np.linalg.norm(I, axis=1, keepdims=True)).astype(np.float32)
    N = rng.normal(size=(64, 3))
    N = (N / np.linalg.norm(N, axis=1, keepdims=True)).astype(np.float32)
    eta = np.where(np.arange(64) % 2 == 0, 1 / 1.5, 1.5).astype(np.float32)
    for i in range(64):
        L.orc_kat_schlick(f32p(f0[i]), f32p(f90[i]), float(c[i]), f32p(sch[i]))
        vs[i] = L.orc_kat_v_smith(float(c[i]), float(xi[i, 0]), float(a[i]))
        L.orc_kat_refract(f32p(I[i]), f32p(N[i]), float(eta[i]), f32p(refr[i]))
    out.update(xi=xi, ggx_a=a, cos_dir=cosd, ggx=ggx, f0=f0, sch_c=c, schlick=sch, v_smith=vs, refr_I=I, refr_N=N,
               refr_eta=eta, refract=refr)
    # colour curves and the sampler
    lin = np.array([L.orc_kat_srgb_to_linear(i / np.float32(255)) for i in range(256)], np.float32)
    xs = np.linspace(0, 1, 257).astype(np.float32)
    enc = np.array([L.orc_kat_linear_to_srgb(float(v)) for v in xs], np.float32)
    keys = np.array([[L.orc_kat_rng_camera_key(s, p, k) for k in range(4)] for s in (0, 1234) for p in (0, 1, 479999)], np.uint32)
    uni = np.array([[L.orc_kat_rng_uniform(int(k), j) for j in range(4)] for k in keys.reshape(-1)], np.float32)
    child = np.array([[L.orc_kat_rng_child(int(k), j) for j in range(6)] for k in keys.reshape(-1)], np.uint32)
    out.update(srgb_lut=lin, srgb_enc_x=xs, srgb_enc=enc, rng_keys=keys, rng_uniform=uni, rng_child=child)
    return out


def on_hit_vectors(model, env, width, height):
    """PBR on_hit on the primary hits of a spread of pixels: radiance + spawned rays."""
    st = H.settings(program=2, spp=1, bounces=3)
    o = H.make_oracle(model, env, st, width, height)
    hits = o.render_hits(width, height)
    m, org = o.get_camera_raw()
    sub = ob.subsample_positions(1)[0]
    recs = []
    mats_seen = set()
    mesh_mat = model.meshes["material_index"]
    for y in range(0, height, 2):
        for x in range(0, width, 2):
            h = hits[y, x]
            key = (int(mesh_mat[h["mesh_index"]]) if h["hit"] else -1, int(h["backface"]))
            if key in mats_seen and len(recs) >= 24:
                continue
            mats_seen.add(key)
            # the camera ray of this pixel, restated in numpy float32 (renderer.cc:17-21, 66-75)
            px, py = np.float32(x) + sub[0], np.float32(y) + sub[1]
            nx = np.float32(px * np.float32(1.0 / width) * np.float32(2) - np.float32(1))
            ny = np.float32(-py * np.float32(1.0 / height) * np.float32(2) + np.float32(1))
            M = m.reshape(4, 4)  # rows = columns of the matrix
            d = (M[0, :3] * nx + M[1, :3] * ny) + (M[2, :3] * np.float32(0) + M[3, :3] * np.float32(1))
            d = (d * (np.float32(1) / np.sqrt(np.float32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2], dtype=np.float32))).astype(np.float32)
            ray = ob.OrcRay((C.c_float * 3)(*org), (C.c_float * 3)(*d), (C.c_float * 3)(1, 1, 1), 3, 0,
                            int(ob.lib().orc_kat_rng_camera_key(1234, y * width + x, 0)))
            hh = abi.Hit(float(h["t"]) if h["hit"] else 0.0, float(h["A"]), float(h["B"]), int(h["object_index"]),
                         int(h["mesh_index"]), int(h["triangle_index"]), int(h["hit"]), int(h["backface"]))
            rad, outs = o.on_hit(ray, hh)
            rec = np.zeros(4 + 4 * 12, np.float32)
            rec[0:3] = rad
            rec[3] = len(outs)
            for k, r in enumerate(outs[:4]):
                rec[4 + 12 * k: 4 + 12 * k + 12] = [*r.origin, *r.direction, *r.payload, r.depth, r.any_hit, 0.0]
            recs.append(np.concatenate([[x, y], rec]))
    return np.asarray(recs, np.float32)


def main():
    out = {}
    for spp in (1, 64, 256, 300, 4096):
        out[f"subsamples_{spp}"] = ob.subsample_positions(spp)
    out.update(kat_tables())

    env = scenes.synthetic_env()
    tiny = scenes.tiny_scene()
    w, h = 64, 36
    o = H.make_oracle(tiny, env, H.settings(program=0, spp=1), w, h)
    m, org = o.get_camera_raw()
    out["tiny_ndc2world"] = m
    out["tiny_origin"] = org
    out["tiny_hits_64x36"] = o.render_hits(w, h)
    out["tiny_raycast_64x36"] = o.render_frame(w, h)[0]
    for mesh in range(len(tiny.meshes)):
        info = o.bvh_info(mesh)
        out[f"tiny_bvh_{mesh}"] = np.array([info.n_nodes, info.n_leaves, info.n_triangles, info.min_depth, info.max_depth,
                                            info.min_tris_in_leaf, info.max_tris_in_leaf, info.checksum & 0xffffffff,
                                            info.checksum >> 32], np.uint64)
    out["tiny_on_hit_pbr"] = on_hit_vectors(tiny, env, w, h)
    ao = H.make_oracle(tiny, env, H.settings(program=1, spp=64, bounces=1, newrays=8), w, h)
    out["tiny_ao_64spp"] = ao.render_frame(w, h)[0]
    pbr = H.make_oracle(tiny, env, H.settings(program=2, spp=64, bounces=4), w, h)
    out["tiny_pbr_64spp"] = pbr.render_frame(w, h)[0]

    # configuration 1 (640x360 raycast, BVH, white environment) on the medium stand-in: digests only
    med = scenes.readme_standin(detail=0.2)
    w1, h1 = 640, 360
    o1 = H.make_oracle(med, None, H.settings(program=0, spp=1), w1, h1)
    hits = o1.render_hits(w1, h1)
    out["c1_hit_digests"] = np.array([digest(hits[f]) for f in ("hit", "object_index", "mesh_index", "triangle_index",
                                                                "backface", "t", "A", "B")])
    out["c1_hit_fraction"] = np.array([hits["hit"].mean()])
    out["c1_image_digest"] = np.array([digest(o1.render_frame(w1, h1)[0])])
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), **out)
    print("wrote oracle_golden.npz with", len(out), "arrays;", os.path.getsize(os.path.join(HERE, "oracle_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
