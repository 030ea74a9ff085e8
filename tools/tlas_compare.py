#!/usr/bin/env python3
"""Frame time of a many-object scene with the instance tree and with the full scan of all bounds (PTGPU_TLAS=linear).
usage: tools/tlas_compare.py [n_objects] [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from pathtracer_amd import scenes
n_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
model, env = scenes.many_objects(n_obj), scenes.synthetic_env()
w, h = 800, 600
for mode in ("tree", "linear"):
    if mode == "linear":
        os.environ["PTGPU_TLAS"] = "linear"
    g = H.make_gpu(model, env, H.settings(program=2, spp=spp, bounces=4, wide=True), w, h)
    os.environ.pop("PTGPU_TLAS", None)
    H.gpu_render(g, w, h)
    _, st = H.gpu_render(g, w, h)
    _, cs = g.render_counted(w, h)
    print(f"{n_obj} objects, {mode:6s}: frame {st.total_ms:8.1f} ms  instance pass {st.extend_ms - st.traverse_ms:8.1f} ms  traverse {st.traverse_ms:7.1f}  "
          f"shade {st.shade_ms:6.1f}  {st.rays / st.total_ms / 1e3:7.1f} Mrays/s  bounds tested per ray {cs.instances_tested / cs.rays:.1f}", flush=True)
    del g
