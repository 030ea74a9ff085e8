"""Small fixed workload for profiling: README stand-in, PBR, 800x600, low spp. Prints per-kernel times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import helpers as H
from pathtracer_amd import scenes, image_io
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prog = int(sys.argv[2]) if len(sys.argv) > 2 else 2
detail = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
wide = (int(sys.argv[5]) if len(sys.argv) > 5 else 1) != 0  # four-wide quantised nodes (what the benchmark runs)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
model = scenes.readme_standin(detail=detail)
env = image_io.load_hdr(os.path.join(root, "assets", "farmland_overcast_1k.hdr"))
w, h = 800, 600
st = H.settings(program=prog, spp=spp, bounces=6 if prog == 2 else 1, newrays=8 if prog == 1 else 0, wide=wide)
g = H.make_gpu(model, env, st, w, h)
for i in range(reps):
    t = time.time(); img, s = H.gpu_render(g, w, h); dt = time.time() - t
    print(f"rep {i}: wall {dt*1e3:.1f} ms total {s.total_ms:.1f} extend {s.extend_ms:.1f} (instance pass {s.extend_ms - s.traverse_ms:.2f}) shade {s.shade_ms:.1f} gen {s.generate_ms:.1f} "
          f"rays {s.rays} launches {s.extend_launches} Mrays/s {s.rays/dt/1e6:.1f} extend-only {s.rays/s.extend_ms/1e3:.1f}", flush=True)

img, c = g.render_counted(w, h)
print(f"counted: V {c.nodes_visited/c.rays:.2f} T {c.triangles_tested/c.rays:.2f} E {c.instances_entered/c.rays:.2f} "
      f"inner util {c.nodes_visited/64/max(c.wave_inner_iterations,1):.3f} tri util {c.triangles_tested/64/max(c.wave_triangle_iterations,1):.3f} "
      f"wave inner iters/wave {c.wave_inner_iterations/(c.rays/64):.1f} tri iters/wave {c.wave_triangle_iterations/(c.rays/64):.1f}")
