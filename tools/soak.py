#!/usr/bin/env python3
"""Repeat one frame many times: ray counts must repeat exactly, pixels within the summation-order tolerance, device
memory must not grow.  usage: tools/soak.py [repeats] [spp]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import helpers as H
from pathtracer_amd import scenes, image_io
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
model = scenes.readme_standin(detail=1.0)
env = image_io.load_hdr(os.path.join(ROOT, "assets", "farmland_overcast_1k.hdr"))
w, h = 800, 600
cases = [(2, True, False), (2, False, False), (2, True, True), (1, True, False), (0, True, False)]  # program, wide nodes, single path
for prog, wide, single in cases:
    g = H.make_gpu(model, env, H.settings(program=prog, spp=spp, bounces=6 if prog == 2 else 1, newrays=8 if prog == 1 else 0,
                                          wide=wide, single=single), w, h)
    first, rays0, free0 = None, None, None
    worst = 0.0
    for i in range(reps):
        img, st = H.gpu_render(g, w, h)
        free = torch.cuda.mem_get_info()[0]
        if first is None:
            first, rays0, free0 = img.astype(np.float64), st.rays, free
        assert st.rays == rays0, (i, st.rays, rays0)
        d = np.abs(img[..., :3] - first[..., :3]) / np.maximum(1.0, np.abs(first[..., :3]))
        worst = max(worst, float(d.max()))
        assert free >= free0 - (64 << 20), (i, free0, free)
    print(f"program={prog} wide={wide} single={single}: {reps} frames, {rays0} rays each, largest relative pixel difference {worst:.2e}, free memory steady", flush=True)
    assert worst <= 2e-5
    del g
