#!/usr/bin/env python3
"""Developer aid: the full 1920x1080 ambient-occlusion frame at 4 spp on the GPU (persistent and one-ray-per-lane kernels)
against the oracle, listing the pixels that differ."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from pathtracer_amd import abi, scenes
model = scenes.readme_standin(detail=1.0)
w, h, r = 1920, 1080, 8
st4 = H.settings(program=1, spp=4, bounces=1, newrays=r)
o = H.make_oracle(model, None, st4, w, h)
full4, so4 = o.render_frame(w, h)
for name, opts in (("persistent", {}), ("simple", dict(flags=abi.OPT_EXTEND_SIMPLE))):
    g = H.make_gpu(model, None, st4, w, h, **opts)
    img4, s4 = H.gpu_render(g, w, h)
    bad = np.argwhere(img4[..., 0] != full4[..., 0])
    print(name, "rays", s4.rays, so4.rays, "differing pixels", len(bad))
    for y, x in bad[:12]:
        print("  ", y, x, img4[y, x, 0] * 32, full4[y, x, 0] * 32)
    if name == "persistent":
        keep = bad
# per-sample check of the first differing pixels: spp=4 is the 2x2 grid; render each sample alone
from pathtracer_amd.distributed import subset_settings
for k in range(4):
    stk = subset_settings(st4, k, 4)
    ok_ = H.make_oracle(model, None, stk, w, h)
    sums, _ = ok_.render_frame(w, h, sums=True)
    g = H.make_gpu(model, None, stk, w, h)
    import torch
    buf = torch.zeros((h, w, 4), device="cuda")
    g.render_frame_sums(w, h, buf.data_ptr())
    gs = buf.cpu().numpy()
    bad = np.argwhere(gs[..., 0] != sums[..., 0])
    print("sample", k, "differing pixels", len(bad), [(int(y), int(x), float(gs[y, x, 0] * 8), float(sums[y, x, 0] * 8)) for y, x in bad[:6]])
