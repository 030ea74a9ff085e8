"""Shared test plumbing: build an oracle scene and a GPU renderer on the same model, camera and settings."""
from __future__ import annotations

import numpy as np

from pathtracer_amd import abi, scenes
from pathtracer_amd.renderer import (AccelerationStructureType, CPUFrameBuffer, Perspective, RayProgramMode,
                                     Renderer, RenderSettings)
from oracle import binding as ob


def settings(program=2, spp=1, bounces=4, newrays=0, as_type=1, leaf=8, seed=1234, env_rot=0.0, begin=0, step=1,
             single=False, wide=False):
    return RenderSettings(spp, bounces, newrays, leaf, env_rot, RayProgramMode(program),
                          AccelerationStructureType(as_type), seed, begin, step, single, wide)


def make_oracle(model, env, st: RenderSettings, width, height, math_mode=ob.MATH_PORTABLE, rng_mode=ob.RNG_COUNTER):
    s = ob.OracleScene(model, env, st.to_abi())
    cam = model.cameras[0]
    s.update_camera_transform_state(cam.position, cam.direction, cam.up, cam.yfov, width / height, cam.znear, cam.zfar)
    s.set_math_mode(math_mode)
    s.set_rng_mode(rng_mode)
    return s


def make_gpu(model, env, st: RenderSettings, width, height, device=0, **options):
    """`options`: fields of the library's tuning record (ptgpu_options) by name, e.g. traverse_waves=6."""
    r = Renderer(device)
    if options:
        r.set_options(**options)
    r.set_render_settings(st)
    r.load_scene(model, env)
    cam = model.cameras[0]
    r.update_camera_transform_state(cam.position, cam.direction, cam.up,
                                    Perspective(cam.yfov, cam.znear, width / height, cam.zfar))
    return r


def gpu_render(r: Renderer, width, height):
    fb = CPUFrameBuffer(width, height)
    r.render_frame(fb)
    return fb.data, r.last_stats


def hits_equal(a, b):
    """Per-pixel equality of two hit dumps: ids exact, t/A/B bit-exact."""
    same = (a["hit"] == b["hit"])
    both = (a["hit"] == 1) & (b["hit"] == 1)
    ids = (a["triangle_index"] == b["triangle_index"]) & (a["object_index"] == b["object_index"]) & \
          (a["mesh_index"] == b["mesh_index"]) & (a["backface"] == b["backface"])
    vals = (a["t"].view(np.uint32) == b["t"].view(np.uint32)) & (a["A"].view(np.uint32) == b["A"].view(np.uint32)) & \
           (a["B"].view(np.uint32) == b["B"].view(np.uint32))
    return same & (~both | (ids & vals)), same & (~both | ids)
