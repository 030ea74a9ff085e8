#!/usr/bin/env python3
"""One rank's share of the README frame at a given world size, timed on one GPU: sample split against tile split.
usage: tools/share_timing.py [world]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import helpers as H
from pathtracer_amd import scenes, image_io
from pathtracer_amd.distributed import subset_settings, subset_tiles
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model = scenes.readme_standin(detail=1.0)
env = image_io.load_hdr(os.path.join(ROOT, "assets", "farmland_overcast_1k.hdr"))
w, h = 800, 600
base = H.settings(program=2, spp=300, bounces=6, wide=True)
g = H.make_gpu(model, env, base, w, h)
def timed(st, label):
    g.set_render_settings(st)
    best = 1e9; rays = 0
    for _ in range(3):
        t = time.perf_counter(); _, s = H.gpu_render(g, w, h); dt = time.perf_counter() - t
        if dt < best: best, keep = dt, s
        rays = s.rays
    s = keep
    print(f"{label}: {best*1e3:.1f} ms, {rays} rays, {rays/best/1e6:.0f} Mrays/s  (device {s.total_ms:.1f}: extend {s.extend_ms:.1f} "
          f"[traverse {s.traverse_ms:.1f}] shade {s.shade_ms:.1f} generate {s.generate_ms:.1f}, {s.extend_launches} launches)", flush=True)
    return best
full = timed(base, "whole frame")
for rank in (0, world - 1):
    a = timed(subset_settings(base, rank, world), f"samples rank {rank}/{world}")
    b = timed(subset_tiles(base, rank, world), f"tiles   rank {rank}/{world}")
    print(f"  ideal {full/world*1e3:.1f} ms; samples {full/world/a:.3f}, tiles {full/world/b:.3f} of ideal", flush=True)
