"""Developer aid: scheduler census of the persistent traversal kernel (lane states per phase) on a bench configuration.
usage: census.py [config] [spp]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ["PTGPU_VERBOSE"] = "1"
import bench
from pathtracer_amd.renderer import AccelerationStructureType, Perspective, RayProgramMode, Renderer, RenderSettings
name = sys.argv[1] if len(sys.argv) > 1 else "readme_pbr"
cfg = dict(bench.CONFIGS[name])
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
model = bench.build_scene(cfg["scene"], 1.0, 10_000_000)
env = bench.load_env(cfg["env"])
w, h = cfg["width"], cfg["height"]
r = Renderer(0)
r.set_render_settings(RenderSettings(spp, cfg["bounces"], cfg["newrays"], 8, 0.0, RayProgramMode(cfg["program"]), AccelerationStructureType.BVH,
                                     seed=1234, singlePath=bool(cfg.get("single")), wideNodes=True))
r.load_scene(model, env)
cam = model.cameras[0]
r.update_camera_transform_state(cam.position, cam.direction, cam.up, Perspective(cam.yfov, cam.znear, w / h, cam.zfar))
_, c = r.render_counted(w, h)
print(f"rays {c.rays} V {c.nodes_visited / c.rays:.2f} (from LDS {c.nodes_from_lds / c.rays:.2f}) T {c.triangles_tested / c.rays:.2f} E {c.instances_entered / c.rays:.2f} "
      f"node-step lane use {c.nodes_visited / 64 / max(c.wave_inner_iterations, 1):.3f} leaf-step lane use (triangle tests per lane-step, up to 2) "
      f"{c.triangles_tested / 64 / max(c.wave_triangle_iterations, 1):.3f} wave node steps per 64 rays {c.wave_inner_iterations / (c.rays / 64):.1f} "
      f"wave leaf steps per 64 rays {c.wave_triangle_iterations / (c.rays / 64):.1f}")
