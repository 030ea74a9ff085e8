#!/usr/bin/env python3
"""Binary nodes against four-wide quantised nodes (PTGPU_FLAG_WIDE_NODES) on one scene: primary-hit agreement, frame time,
visits per ray.  usage: tools/wide_compare.py readme|stress [n_triangles] [spp]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from pathtracer_amd import scenes, image_io

kind = sys.argv[1] if len(sys.argv) > 1 else "readme"
n_tris = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
if kind == "readme":
    model = scenes.readme_standin(detail=1.0)
    env = image_io.load_hdr(os.path.join(ROOT, "assets", "farmland_overcast_1k.hdr"))
    w, h, bounces = 800, 600, 6
else:
    model, env = scenes.stress_mesh(n_tris), None
    w, h, bounces = 3840, 2160, 8
ref_hits = None
for wide in (0, 1):
    g = H.make_gpu(model, env, H.settings(program=2, spp=spp, bounces=bounces, wide=bool(wide)), w, h)
    hits = g.render_hits(w, h)
    if ref_hits is None:
        ref_hits = hits
    exact, ids = H.hits_equal(ref_hits, hits)
    H.gpu_render(g, w, h)
    img, st = H.gpu_render(g, w, h)
    _, cs = g.render_counted(w, h)
    print(f"wide={wide} frame {st.total_ms:8.1f} ms extend {st.extend_ms:8.1f} traverse {st.traverse_ms:8.1f} shade {st.shade_ms:7.1f}  "
          f"{st.rays / st.total_ms / 1e3:7.1f} Mrays/s  rays {st.rays}  nodes/ray {cs.nodes_visited / cs.rays:.2f} tris/ray "
          f"{cs.triangles_tested / cs.rays:.2f}  primary hits identical {exact.mean():.6f} (ids {ids.mean():.6f}) mean {img[..., :3].mean():.6f}",
          flush=True)
    del g
