"""ctypes mirror of ``include/ptgpu.h`` (struct layouts, error codes, prototypes).

The structs describe the scene data contract of the reference
(``src/model_loader.h:14-65``) as flat arrays; the same layouts are accepted
by the CPU oracle, so one set of numpy buffers feeds both.
"""
from __future__ import annotations

import ctypes as C

PTGPU_OK = 0
PROGRAM_RAYCASTER, PROGRAM_AO, PROGRAM_PBR = 0, 1, 2
AS_NAIVE, AS_BVH = 0, 1

c_float_p = C.POINTER(C.c_float)
c_u32_p = C.POINTER(C.c_uint32)
c_u8_p = C.POINTER(C.c_uint8)


class Vertex(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("normal", C.c_float * 3), ("tangent", C.c_float * 4),
                ("uv", C.c_float * 2)]


class Mesh(C.Structure):
    _fields_ = [("first_vertex", C.c_uint32), ("n_vertices", C.c_uint32), ("first_index", C.c_uint32),
                ("n_indices", C.c_uint32), ("material_index", C.c_uint32)]


class Object(C.Structure):
    _fields_ = [("model_matrix", C.c_float * 16), ("normal_matrix", C.c_float * 16), ("mesh_index", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [
        ("base_color_factor", C.c_float * 4),
        ("emissive_factor", C.c_float * 3),
        ("attenuation_factor", C.c_float * 3),
        ("metallic_factor", C.c_float),
        ("roughness_factor", C.c_float),
        ("base_color_texture", C.c_int32),
        ("metallic_roughness_texture", C.c_int32),
        ("normal_texture", C.c_int32),
        ("ior", C.c_float),
        ("dielectric_f0", C.c_float),
        ("transmission_factor", C.c_float),
        ("transmission_texture", C.c_int32),
        ("emissive_texture", C.c_int32),
        ("emissive_strength", C.c_float),
        ("double_sided", C.c_int32),
        ("has_volume", C.c_int32),
        ("alpha_blending", C.c_int32),
        ("alpha_cutoff", C.c_float),
    ]


class Image(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("texel_offset", C.c_uint64)]


class Model(C.Structure):
    _fields_ = [
        ("vertices", C.POINTER(Vertex)), ("n_vertices", C.c_uint64),
        ("indices", c_u32_p), ("n_indices", C.c_uint64),
        ("meshes", C.POINTER(Mesh)), ("n_meshes", C.c_uint32),
        ("objects", C.POINTER(Object)), ("n_objects", C.c_uint32),
        ("materials", C.POINTER(Material)), ("n_materials", C.c_uint32),
        ("images", C.POINTER(Image)), ("n_images", C.c_uint32),
        ("texels", c_u8_p), ("n_texels", C.c_uint64),
    ]


class Envmap(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgba", c_float_p)]


class Settings(C.Structure):
    _fields_ = [
        ("samples_per_pixel", C.c_uint32),
        ("max_ray_bounces", C.c_uint32),
        ("max_new_rays_per_bounce", C.c_uint32),
        ("max_triangles_per_bvh_leaf", C.c_uint32),
        ("envmap_rotation", C.c_float),
        ("program_mode", C.c_int32),
        ("accel_struct_type", C.c_int32),
        ("seed", C.c_uint32),
        ("sample_begin", C.c_uint32),
        ("sample_step", C.c_uint32),
        ("flags", C.c_uint32),
        ("tile_begin", C.c_uint32),
        ("tile_step", C.c_uint32),
    ]


FLAG_SINGLE_PATH = 1
FLAG_WIDE_NODES = 2


class Hit(C.Structure):
    _fields_ = [("t", C.c_float), ("A", C.c_float), ("B", C.c_float), ("object_index", C.c_uint32),
                ("mesh_index", C.c_uint32), ("triangle_index", C.c_uint32), ("hit", C.c_uint32),
                ("backface", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64), ("camera_samples", C.c_uint64), ("extend_launches", C.c_uint64),
        ("shade_launches", C.c_uint64), ("extend_ms", C.c_double), ("shade_ms", C.c_double),
        ("generate_ms", C.c_double), ("total_ms", C.c_double), ("nodes_visited", C.c_uint64),
        ("triangles_tested", C.c_uint64), ("instances_tested", C.c_uint64), ("instances_entered", C.c_uint64),
        ("wave_inner_iterations", C.c_uint64), ("wave_triangle_iterations", C.c_uint64),
        ("traverse_ms", C.c_double), ("nodes_from_lds", C.c_uint64), ("traverse_workgroups_per_cu", C.c_uint32), ("ragged_end_in_pieces", C.c_uint32),
        ("leaf_records_read", C.c_uint64), ("rays_abandoned", C.c_uint64),
    ]


class Options(C.Structure):
    """``ptgpu_options``: tuning of a context; zero fields mean the library's choice."""
    _fields_ = [("chunk_rays", C.c_uint32), ("pool_gib", C.c_uint32), ("traverse_waves", C.c_uint32), ("leaf_vote", C.c_uint32),
                ("build_threads", C.c_uint32), ("intersect_chunk", C.c_uint32), ("persist_blocks", C.c_uint32), ("flags", C.c_uint32)]


OPT_EXTEND_SIMPLE, OPT_REFERENCE_SPLITS, OPT_GREEDY_COLLAPSE, OPT_LINEAR_TLAS, OPT_VERBOSE, OPT_VERBOSE_LAUNCHES = 1, 2, 4, 8, 16, 32
OPT_LEAF_RECORDS, OPT_NO_LEAF_RECORDS, OPT_ONE_COUNTER, OPT_SMALL_STAGING = 64, 128, 256, 512


class BvhInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_leaves", C.c_uint32), ("n_triangles", C.c_uint32),
                ("min_depth", C.c_uint32), ("max_depth", C.c_uint32), ("mean_depth", C.c_float),
                ("min_tris_in_leaf", C.c_uint32), ("max_tris_in_leaf", C.c_uint32), ("mean_tris_in_leaf", C.c_float),
                ("checksum", C.c_uint64)]


# numpy dtypes with the same layout (used for bulk I/O of hits and vertices)
import numpy as np  # noqa: E402

HIT_DTYPE = np.dtype([("t", "<f4"), ("A", "<f4"), ("B", "<f4"), ("object_index", "<u4"), ("mesh_index", "<u4"),
                      ("triangle_index", "<u4"), ("hit", "<u4"), ("backface", "<u4")])
VERTEX_DTYPE = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3), ("tangent", "<f4", 4), ("uv", "<f4", 2)])
MESH_DTYPE = np.dtype([("first_vertex", "<u4"), ("n_vertices", "<u4"), ("first_index", "<u4"), ("n_indices", "<u4"),
                       ("material_index", "<u4")])
OBJECT_DTYPE = np.dtype([("model_matrix", "<f4", 16), ("normal_matrix", "<f4", 16), ("mesh_index", "<u4")])
IMAGE_DTYPE = np.dtype([("width", "<i4"), ("height", "<i4"), ("texel_offset", "<u8")])
MATERIAL_DTYPE = np.dtype([
    ("base_color_factor", "<f4", 4), ("emissive_factor", "<f4", 3), ("attenuation_factor", "<f4", 3),
    ("metallic_factor", "<f4"), ("roughness_factor", "<f4"), ("base_color_texture", "<i4"),
    ("metallic_roughness_texture", "<i4"), ("normal_texture", "<i4"), ("ior", "<f4"), ("dielectric_f0", "<f4"),
    ("transmission_factor", "<f4"), ("transmission_texture", "<i4"), ("emissive_texture", "<i4"),
    ("emissive_strength", "<f4"), ("double_sided", "<i4"), ("has_volume", "<i4"), ("alpha_blending", "<i4"),
    ("alpha_cutoff", "<f4")])

assert HIT_DTYPE.itemsize == C.sizeof(Hit)
assert VERTEX_DTYPE.itemsize == C.sizeof(Vertex) == 48
assert MESH_DTYPE.itemsize == C.sizeof(Mesh)
assert OBJECT_DTYPE.itemsize == C.sizeof(Object)
assert IMAGE_DTYPE.itemsize == C.sizeof(Image)
assert MATERIAL_DTYPE.itemsize == C.sizeof(Material)


def declare_ptgpu(lib: C.CDLL) -> None:
    """Attach argument/return types for every entry point include/ptgpu.h declares."""
    vp = C.c_void_p
    P = C.POINTER
    sig = {
        "ptgpu_create": (C.c_int, [C.c_int, P(vp)]),
        "ptgpu_destroy": (None, [vp]),
        "ptgpu_last_error": (C.c_char_p, [vp]),
        "ptgpu_set_render_settings": (C.c_int, [vp, P(Settings)]),
        "ptgpu_get_render_settings": (C.c_int, [vp, P(Settings)]),
        "ptgpu_set_options": (C.c_int, [vp, P(Options)]),
        "ptgpu_get_options": (C.c_int, [vp, P(Options)]),
        "ptgpu_load_scene": (C.c_int, [vp, P(Model), P(Envmap)]),
        "ptgpu_load_scene_from": (C.c_int, [vp, vp]),
        "ptgpu_host_build_count": (C.c_uint64, []),
        "ptgpu_update_camera_transform_state": (C.c_int, [vp, c_float_p, c_float_p, c_float_p, C.c_float, C.c_float,
                                                          C.c_float, C.c_float]),
        "ptgpu_set_camera_raw": (C.c_int, [vp, c_float_p, c_float_p]),
        "ptgpu_get_camera_raw": (C.c_int, [vp, c_float_p, c_float_p]),
        "ptgpu_get_scene_bound": (C.c_int, [vp, c_float_p, c_float_p]),
        "ptgpu_render_frame": (C.c_int, [vp, C.c_int, C.c_int, c_float_p, P(Stats)]),
        "ptgpu_render_frame_sums": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, P(Stats)]),
        "ptgpu_resolve_sums": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_uint32, vp]),
        "ptgpu_render_counted": (C.c_int, [vp, C.c_int, C.c_int, c_float_p, P(Stats)]),
        "ptgpu_render_hits": (C.c_int, [vp, C.c_int, C.c_int, C.c_uint32, P(Hit)]),
        "ptgpu_intersect_rays": (C.c_int, [vp, C.c_uint64, c_float_p, c_float_p, c_u8_p, P(Hit)]),
        "ptgpu_intersect_rays_static": (C.c_int, [vp, C.c_uint32, vp, vp, vp, vp, vp, C.c_int]),  # device pointers
        "ptgpu_get_bvh_info": (C.c_int, [vp, C.c_uint32, P(BvhInfo)]),
        "ptgpu_subsample_positions": (C.c_int, [C.c_uint32, c_float_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args


STATIC_MAX_RAYS = 1 << 17  # PTGPU_STATIC_MAX_RAYS

PTGPU_SYMBOLS = [
    "ptgpu_create", "ptgpu_destroy", "ptgpu_last_error", "ptgpu_set_render_settings", "ptgpu_get_render_settings",
    "ptgpu_load_scene", "ptgpu_update_camera_transform_state", "ptgpu_set_camera_raw", "ptgpu_get_camera_raw",
    "ptgpu_get_scene_bound", "ptgpu_render_frame", "ptgpu_render_frame_sums", "ptgpu_resolve_sums",
    "ptgpu_render_counted", "ptgpu_render_hits", "ptgpu_intersect_rays", "ptgpu_intersect_rays_static", "ptgpu_get_bvh_info",
    "ptgpu_subsample_positions", "ptgpu_set_options", "ptgpu_get_options", "ptgpu_load_scene_from", "ptgpu_host_build_count",
]
