import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import helpers as H
from pathtracer_amd import scenes
from oracle import binding as ob
sc = scenes.readme_standin(detail=0.08)
env = scenes.synthetic_env()
print("tris", sc.n_triangles, flush=True)
W, Hh = 160, 90
st = H.settings(program=0, spp=1)
o = H.make_oracle(sc, env, st, W, Hh)
g = H.make_gpu(sc, env, st, W, Hh)
print("camera equal:", np.array_equal(o.get_camera_raw()[0], g.get_camera_raw()[0]), flush=True)
for m in range(len(sc.meshes)):
    a, b = o.bvh_info(m), g.bvh_info(m)
    print("mesh", m, a.n_nodes, b.n_nodes, a.checksum == b.checksum, a.max_depth, b.max_depth)
ho = o.render_hits(W, Hh); hg = g.render_hits(W, Hh)
full, ids = H.hits_equal(ho, hg)
print("hits exact", full.mean(), "ids", ids.mean(), "hit frac", ho['hit'].mean(), flush=True)
io, _ = o.render_frame(W, Hh); ig, s = H.gpu_render(g, W, Hh)
print("raycast img maxdiff", np.abs(io-ig).max(), "exact", np.array_equal(io, ig), flush=True)
for prog, spp, b, r in ((1, 4, 1, 8), (2, 4, 4, 0)):
    st = H.settings(program=prog, spp=spp, bounces=b, newrays=r)
    o = H.make_oracle(sc, env, st, W, Hh); g = H.make_gpu(sc, env, st, W, Hh)
    io, so = o.render_frame(W, Hh); ig, sg = H.gpu_render(g, W, Hh)
    d = np.abs(io[..., :3]-ig[..., :3])
    print("prog", prog, "rays", so.rays, sg.rays, "maxdiff", d.max(), "meandiff", d.mean(), "frac>1e-4", (d.max(-1) > 1e-4).mean(),
          "mean", io[..., :3].mean(), ig[..., :3].mean(), "gpu ms", sg.total_ms, "extend ms", sg.extend_ms, flush=True)
