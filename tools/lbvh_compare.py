#!/usr/bin/env python3
"""Scene-load time and rendering rate of the two tree builders (host binned SAH / device linear BVH) on one scene.
usage: tools/lbvh_compare.py readme|stress [n_triangles] [spp]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import helpers as H
from pathtracer_amd import scenes, image_io

kind = sys.argv[1] if len(sys.argv) > 1 else "stress"
n_tris = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 8
if kind == "readme":
    model = scenes.readme_standin(detail=1.0)
    env = image_io.load_hdr(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "farmland_overcast_1k.hdr"))
    w, h, bounces = 800, 600, 6
else:
    model, env = scenes.stress_mesh(n_tris), None
    w, h, bounces = 3840, 2160, 8
print(f"{kind}: {model.instanced_triangles()} instanced triangles, {w}x{h}, {spp} spp, {bounces} bounces", flush=True)
for as_type, name in ((1, "host SAH"), (2, "device LBVH")):
    t0 = time.time()
    g = H.make_gpu(model, env, H.settings(program=2, spp=spp, bounces=bounces, as_type=as_type), w, h)
    t_load = time.time() - t0
    H.gpu_render(g, w, h)
    _, st = H.gpu_render(g, w, h)
    _, cs = g.render_counted(w, h)
    info = g.bvh_info(0)
    print(f"{name:12s} load {t_load:6.2f} s  frame {st.total_ms:8.1f} ms  {st.rays / st.total_ms / 1e3:7.1f} Mrays/s  "
          f"nodes/ray {cs.nodes_visited / cs.rays:.2f} tris/ray {cs.triangles_tested / cs.rays:.2f}  "
          f"mesh 0: {info.n_nodes} nodes, depth {info.max_depth}, {info.mean_tris_in_leaf:.2f} tris/leaf", flush=True)
    del g
