#!/usr/bin/env python3
"""The README frame at full sample count over binary and over four-wide nodes: how many pixels differ beyond the
summation-order tolerance, by how much, and how many rays each traced."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from pathtracer_amd import scenes, image_io
model = scenes.readme_standin(detail=1.0)
env = image_io.load_hdr(os.path.join(ROOT, "assets", "farmland_overcast_1k.hdr"))
w, h, spp = 800, 600, int(sys.argv[1]) if len(sys.argv) > 1 else 300
imgs, rays = [], []
for wide in (False, True):
    g = H.make_gpu(model, env, H.settings(program=2, spp=spp, bounces=6, wide=wide), w, h)
    img, st = H.gpu_render(g, w, h)
    imgs.append(img[..., :3].astype(np.float64)); rays.append(st.rays)
    print(f"wide={wide}: {st.rays} rays, {st.total_ms:.1f} ms", flush=True)
    del g
d = np.abs(imgs[0] - imgs[1])
tol = 2e-5 * np.maximum(1.0, np.abs(imgs[0]))
bad = (d > tol).any(axis=-1)
print(f"ray count difference {rays[1] - rays[0]}; pixels beyond tolerance {bad.sum()} of {w * h}; max |diff| {d.max():.3e}; "
      f"mean |diff| {d.mean():.3e}; image mean {imgs[0].mean():.6f} / {imgs[1].mean():.6f}")
