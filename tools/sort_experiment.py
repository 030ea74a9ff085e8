"""Developer experiment: does ordering the rays of one launch by (direction octant, origin cell) speed the
traversal kernel up?  Usage: sort_experiment.py dump [launch] | plain | sorted [od|do]"""
import os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
mode = sys.argv[1]
path = "/tmp/rays.bin"
if mode == "dump":
    os.environ["PTGPU_DUMP_RAYS"] = path
    os.environ["PTGPU_DUMP_LAUNCH"] = sys.argv[2] if len(sys.argv) > 2 else "12"
import helpers as H
from pathtracer_amd import scenes, image_io
stress = os.environ.get("SCENE", "readme") == "stress"
model = scenes.stress_mesh(10_000_000) if stress else scenes.readme_standin(detail=1.0)
env = None if stress else image_io.load_hdr(os.path.join(root, "assets", "farmland_overcast_1k.hdr"))
W, Hh = (3840, 2160) if stress else (800, 600)
wide = os.environ.get("WIDE", "1") != "0"
g = H.make_gpu(model, env, H.settings(program=2, spp=int(os.environ.get("SPP", "8")), bounces=8 if stress else 6, newrays=0, wide=wide), W, Hh)
if mode == "dump":
    H.gpu_render(g, W, Hh)
    print("dumped", os.path.getsize(path))
    sys.exit(0)
raw = np.fromfile(path, dtype=np.float32).reshape(2, -1, 4)
limit = int(os.environ.get("MAX_RAYS", "0")) or raw.shape[1]
o = np.ascontiguousarray(raw[0, :limit, :3]); d = np.ascontiguousarray(raw[1, :limit, :3])
del raw
n = o.shape[0]
if mode == "sorted":
    lo = o.min(0); hi = o.max(0)
    order = sys.argv[2] if len(sys.argv) > 2 else "od"
    bits = 10 if order == "fine" else (int(order[4:]) if order.startswith("bits") else 4)
    if order.startswith("local"): bits = 4
    if order == "hit":  # cells of the point the ray came from are replaced by where it goes: origin + direction * (scene size / 4)
        o_key = o + d * float(np.linalg.norm(hi - lo)) * 0.25
    else:
        o_key = o
    cell = np.clip(((o_key - lo) / np.maximum(hi - lo, 1e-9) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    def spread(v):
        r = np.zeros_like(v)
        for b in range(bits):
            r |= ((v >> b) & 1) << (3 * b)
        return r
    morton = spread(cell[:, 0]) | (spread(cell[:, 1]) << 1) | (spread(cell[:, 2]) << 2)
    octant = (d[:, 0] < 0).astype(np.int64) | ((d[:, 1] < 0).astype(np.int64) << 1) | ((d[:, 2] < 0).astype(np.int64) << 2)
    key = ((octant << 12) | morton) if order == "od" else (morton if order.startswith("bits") else ((morton << 3) | octant))
    if order.startswith("local"):  # reorder inside windows of W consecutive rays only: by octant ("localW") or octant + fine cell ("localfullW")
        full = order.startswith("localfull")
        W = int(order[9:] if full else order[5:])
        window = np.arange(n, dtype=np.int64) // W
        if full:
            cell10 = np.clip(((o - lo) / np.maximum(hi - lo, 1e-9) * 1024).astype(np.int64), 0, 1023)
            def spread10(v):
                r = np.zeros_like(v)
                for b_ in range(10):
                    r |= ((v >> b_) & 1) << (3 * b_)
                return r
            m30 = spread10(cell10[:, 0]) | (spread10(cell10[:, 1]) << 1) | (spread10(cell10[:, 2]) << 2)
            key = (window << 33) | (octant << 30) | m30
        else:
            key = (window << 3) | octant
    perm = np.argsort(key, kind="stable")
    o = np.ascontiguousarray(o[perm]); d = np.ascontiguousarray(d[perm])
for rep in range(3):
    t0 = time.perf_counter()
    hits = g.intersect_rays(o, d)
    print(mode, sys.argv[2:], "rays", n, "wall ms", round((time.perf_counter() - t0) * 1e3, 1), "hit frac", float((hits["hit"] != 0).mean()), flush=True)
