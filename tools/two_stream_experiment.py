#!/usr/bin/env python3
"""Developer experiment: do two contexts rendering complementary sample subsets concurrently on one GPU (host threads,
separate streams) finish the frame sooner than one context rendering all samples?  usage: two_stream_experiment.py [spp] [ways]"""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import helpers as H
from pathtracer_amd import scenes, image_io
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ways = int(sys.argv[2]) if len(sys.argv) > 2 else 2
model = scenes.readme_standin(detail=1.0)
env = image_io.load_hdr(os.path.join(ROOT, "assets", "farmland_overcast_1k.hdr"))
w, h = 800, 600
one = H.make_gpu(model, env, H.settings(program=2, spp=spp, bounces=6, wide=True), w, h)
for rep in range(3):
    t = time.perf_counter(); img, st = H.gpu_render(one, w, h); dt = time.perf_counter() - t
    print(f"one context: {dt*1e3:.1f} ms, {st.rays/dt/1e6:.0f} Mrays/s", flush=True)
ctxs = [H.make_gpu(model, env, H.settings(program=2, spp=spp, bounces=6, wide=True, begin=i, step=ways), w, h) for i in range(ways)]
res = [None] * ways
def work(i):
    res[i] = H.gpu_render(ctxs[i], w, h)
for rep in range(3):
    th = [threading.Thread(target=work, args=(i,)) for i in range(ways)]
    t = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t
    rays = sum(r[1].rays for r in res)
    print(f"{ways} contexts: {dt*1e3:.1f} ms, {rays/dt/1e6:.0f} Mrays/s  (each: {[round(r[1].total_ms,1) for r in res]})", flush=True)
