#!/usr/bin/env python3
"""Frame time of the README stand-in against the leaf size of the trees, binary and four-wide nodes.
usage: tools/leaf_sweep.py [spp] [readme|stress] [n_triangles]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from pathtracer_amd import scenes, image_io

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
if len(sys.argv) > 2 and sys.argv[2] == "stress":
    model, env = scenes.stress_mesh(int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000), None
    w, h, bounces = 3840, 2160, 8
else:
    model = scenes.readme_standin(detail=1.0)
    env = image_io.load_hdr(os.path.join(ROOT, "assets", "farmland_overcast_1k.hdr"))
    w, h, bounces = 800, 600, 6
import time
for wide in (True, False):
    for leaf in (8, 4, 2):
        t0 = time.time()
        g = H.make_gpu(model, env, H.settings(program=2, spp=spp, bounces=bounces, wide=wide, leaf=leaf), w, h)
        H.gpu_render(g, w, h)
        _, st = H.gpu_render(g, w, h)
        _, cs = g.render_counted(w, h)
        print(f"load {time.time() - t0:5.1f} s  wide={int(wide)} leaf<={leaf}: frame {st.total_ms:7.1f} ms traverse {st.traverse_ms:7.1f} extend {st.extend_ms:7.1f} shade {st.shade_ms:6.1f} "
              f"{st.rays / st.total_ms / 1e3:7.1f} Mrays/s nodes/ray {cs.nodes_visited / cs.rays:6.2f} tris/ray {cs.triangles_tested / cs.rays:6.2f} rays {st.rays}",
              flush=True)
        del g
