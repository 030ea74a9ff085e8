/*
 * ptgpu.h — C ABI of libptgpu, the MI355X (gfx950) implementation of the
 * PathTracer rendering hot path.
 *
 * The seam is the reference's `Renderer` class at frame granularity
 * (reference src/renderer.h:37-71).  The per-ray virtual interfaces
 * `IAccelerationStructure::intersect_ray` (src/acceleration_structure.h:63-69)
 * and `IRayProgram::on_hit` (src/ray_program.h:58-63) are selected by enums in
 * `Renderer::load_scene` (src/renderer.cc:44-63); here they become device
 * kernels chosen by the same enums in `ptgpu_settings`.
 *
 * Conventions
 *   - plain C, plain pointers and sizes, no C++ or torch types;
 *   - every call returns 0 on success or a negative PTGPU_E_* code; the text
 *     of the last failure of a context is `ptgpu_last_error(ctx)`;
 *   - matrices are column-major float[16] (m[col*4+row]), as glm stores them;
 *   - host pointers are copied during the call (the caller may free them
 *     afterwards); pointers whose name starts with `d_` are device pointers;
 *   - one context per GPU per process, driven from one host thread.
 *   - the library never falls back to the CPU: without a usable gfx950
 *     device `ptgpu_create` fails.
 */
#ifndef PTGPU_H
#define PTGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes ------------------------------------------------------- */
#define PTGPU_OK 0
#define PTGPU_E_INVALID (-1)   /* bad argument / inconsistent model          */
#define PTGPU_E_NODEVICE (-2)  /* no usable HIP device                       */
#define PTGPU_E_HIP (-3)       /* a HIP runtime call failed                  */
#define PTGPU_E_STATE (-4)     /* call order: scene or camera not set        */
#define PTGPU_E_NOMEM (-5)     /* allocation failed / ray pool too small     */

/* ---- enums (reference src/arguments.h:10-23) --------------------------- */
enum { PTGPU_PROGRAM_RAYCASTER = 0, PTGPU_PROGRAM_AO = 1, PTGPU_PROGRAM_PBR = 2 };
/* PTGPU_AS_LBVH has no counterpart in the reference: the same two-level structure with the per-mesh trees built
 * on the GPU (Morton-ordered linear BVH) instead of by the host's binned-SAH builder — a much faster scene load for
 * multi-million-triangle meshes, a somewhat slower traversal.  Closest hits are those of PTGPU_AS_BVH except
 * where two triangles tie in distance. */
enum { PTGPU_AS_NAIVE = 0, PTGPU_AS_BVH = 1, PTGPU_AS_LBVH = 2 };

/* ---- scene data contract (reference src/model_loader.h:14-65) ---------- */

/* `vertex` (model_loader.h:14-19): 12 floats, 48 bytes. */
typedef struct ptgpu_vertex {
    float position[3];
    float normal[3];
    float tangent[4]; /* xyz + handedness sign in w */
    float uv[2];
} ptgpu_vertex;

/* `Mesh` (model_loader.h:21-25): a glTF primitive.  Indices are relative to
 * the mesh's own vertex range and n_indices is a multiple of 3. */
typedef struct ptgpu_mesh {
    uint32_t first_vertex; /* into ptgpu_model.vertices */
    uint32_t n_vertices;
    uint32_t first_index; /* into ptgpu_model.indices  */
    uint32_t n_indices;
    uint32_t material_index;
} ptgpu_mesh;

/* `Object` (model_loader.h:27-31): NormalMatrix = transpose(inverse(Model)). */
typedef struct ptgpu_object {
    float model_matrix[16];
    float normal_matrix[16];
    uint32_t mesh_index;
} ptgpu_object;

/* `Material` (model_loader.h:33-52); texture indices are image indices, -1 = none. */
typedef struct ptgpu_material {
    float base_color_factor[4];
    float emissive_factor[3];
    float attenuation_factor[3];
    float metallic_factor;
    float roughness_factor;
    int32_t base_color_texture;
    int32_t metallic_roughness_texture;
    int32_t normal_texture;
    float ior;
    float dielectric_f0;
    float transmission_factor;
    int32_t transmission_texture;
    int32_t emissive_texture;
    float emissive_strength;
    int32_t double_sided;
    int32_t has_volume;
    int32_t alpha_blending;
    float alpha_cutoff;
} ptgpu_material;

/* `CPUTexture<sdr_pixel>` (cpu_framebuffer.h:61-133): RGBA8 rows, top row first. */
typedef struct ptgpu_image {
    int32_t width;
    int32_t height;
    uint64_t texel_offset; /* in pixels, into ptgpu_model.texels (4 bytes per pixel) */
} ptgpu_image;

/* `Model` (model_loader.h:59-65) as flat arrays. */
typedef struct ptgpu_model {
    const ptgpu_vertex* vertices;
    uint64_t n_vertices;
    const uint32_t* indices;
    uint64_t n_indices;
    const ptgpu_mesh* meshes;
    uint32_t n_meshes;
    const ptgpu_object* objects;
    uint32_t n_objects;
    const ptgpu_material* materials;
    uint32_t n_materials;
    const ptgpu_image* images;
    uint32_t n_images;
    const uint8_t* texels;
    uint64_t n_texels; /* pixels */
} ptgpu_model;

/* `CPUTexture<hdr_pixel>` environment map: RGBA32F rows, top row first. */
typedef struct ptgpu_envmap {
    int32_t width;
    int32_t height;
    const float* rgba;
} ptgpu_envmap;

/* `RenderSettings` (reference src/render_settings.h:7-21) plus what the GPU
 * path adds: the seed of the counter-based sampler (the reference seeds a
 * thread-local minstd_rand from std::random_device, ray_program.cc:169,231)
 * and the sample subset rendered by this context (multi-GPU sample split). */
typedef struct ptgpu_settings {
    uint32_t samples_per_pixel;
    uint32_t max_ray_bounces; /* used modulo 256 like the uint8 cast at renderer.cc:73 */
    uint32_t max_new_rays_per_bounce;
    uint32_t max_triangles_per_bvh_leaf;
    float envmap_rotation; /* radians */
    int32_t program_mode;  /* PTGPU_PROGRAM_* */
    int32_t accel_struct_type; /* PTGPU_AS_* */
    uint32_t seed;
    uint32_t sample_begin; /* first global sample index rendered here (default 0) */
    uint32_t sample_step;  /* stride between rendered sample indices (default 1; 0 means 1) */
    uint32_t flags;        /* PTGPU_FLAG_*, default 0 */
    uint32_t tile_begin;   /* multi-GPU tile split: first 8x8-pixel tile (row-major tile index) rendered here (default 0) */
    uint32_t tile_step;    /* stride between rendered tiles (default 1; 0 means 1); pixels of other tiles stay 0 */
} ptgpu_settings;

/* PBR only, no counterpart in the reference (whose estimator follows every lobe
 * of every hit, src/ray_program.cc:265-388): of the rays a hit would spawn, one
 * is kept, chosen with probability proportional to the largest component of its
 * payload, and its payload is divided by that probability.  Same expectation,
 * at most one ray per hit, so deep bounce limits stay affordable. */
#define PTGPU_FLAG_SINGLE_PATH 1u

/* Read by ptgpu_load_scene; no counterpart in the reference.  The persistent traversal kernel walks a four-wide,
 * quantised form of the same trees (inner children pulled up into their parent; child boxes stored with 8 bits per
 * plane inside the parent's box, rounded outwards; 64 bytes for four children): about half the node visits and half
 * the node bytes per ray.  The closest hit does not depend on the order in which boxes are visited, so hit records,
 * ray counts and images are those of the binary tree except where two triangles tie in distance; what does depend on
 * the order — WHICH triangle an any-hit ray reports, and the node-visit counters — differs from the reference.
 * Trees for this layout are built with leaves of at most 2 triangles (min(max_triangles_per_bvh_leaf, 2)): fewer
 * triangle tests per ray.  Ignored with PTGPU_AS_NAIVE. */
#define PTGPU_FLAG_WIDE_NODES 2u

/* Tuning of a context; no counterpart in the reference.  Zero in a field means "the library's choice", which is what
 * every measurement in this repository was taken with unless it says otherwise.  A new context starts from the
 * PTGPU_* environment variables of the same names (PTGPU_CHUNK, PTGPU_POOL_GIB, PTGPU_TRAVERSE_WAVES, PTGPU_LEAF_VOTE,
 * PTGPU_BUILD_THREADS, PTGPU_INTERSECT_CHUNK, PTGPU_PERSIST_BLOCKS, PTGPU_EXTEND=simple, PTGPU_WIDE_SPLITS=reference,
 * PTGPU_WIDE_COLLAPSE=greedy, PTGPU_TLAS=linear, PTGPU_LEAF_RECORDS=0|1, PTGPU_RESERVE=one-counter, PTGPU_VERBOSE=1|2), read once in ptgpu_create so that measurement
 * scripts can alternate settings around an unchanged caller; ptgpu_set_options replaces them. */
typedef struct ptgpu_options {
    uint32_t chunk_rays;      /* camera samples per batch, 1024 .. 2^28 (default 2^26): sizes the ray pools and the launches */
    uint32_t pool_gib;        /* upper bound of each of the two ray pools in GiB, 1 .. 256 (default 32) */
    uint32_t traverse_waves;  /* four-wide traversal kernel at 6 or 7 workgroups per CU (default: by the size of the geometry) */
    uint32_t leaf_vote;       /* 1 .. 16: weight of lanes at leaves when the traversal kernel picks its phase (default: by scene) */
    uint32_t build_threads;   /* host threads of the tree build, 1 .. 64 (default: one per core, at most 16) */
    uint32_t intersect_chunk; /* rays per pass of ptgpu_intersect_rays (default 4 Mi) */
    uint32_t persist_blocks;  /* resident workgroups of the traversal kernel (default: occupancy x compute units) */
    uint32_t flags;           /* PTGPU_OPT_* */
} ptgpu_options;
#define PTGPU_OPT_EXTEND_SIMPLE 1u      /* one ray per lane instead of the persistent traversal kernel (cross-check) */
#define PTGPU_OPT_REFERENCE_SPLITS 2u   /* four-wide nodes collapsed from the reference's tree instead of the tuned one */
#define PTGPU_OPT_GREEDY_COLLAPSE 4u    /* four-wide collapse: pull up the largest child instead of the optimal choice */
#define PTGPU_OPT_LINEAR_TLAS 8u        /* every ray tests every instance bound, however many instances */
#define PTGPU_OPT_VERBOSE 16u           /* diagnostics on stderr */
#define PTGPU_OPT_VERBOSE_LAUNCHES 32u  /* one line per launch on stderr */
/* Leaves of the four-wide trees as one aligned 64-byte record each (one memory sector per leaf visit instead of two or
 * three) — by default for scenes whose geometry is beyond the caches; these two force it on or off (PTGPU_LEAF_RECORDS=1|0).
 * The same hits to the bit either way. */
#define PTGPU_OPT_LEAF_RECORDS 64u
#define PTGPU_OPT_NO_LEAF_RECORDS 128u
/* The PBR shading kernel reserves its output slots with one atomic per workgroup on ONE shared counter
 * (PTGPU_RESERVE=one-counter) instead of eight counters feeding interleaved chunks of the output: the older form, which runs
 * at the pace of that one counter word; kept as cross-check.  Same rays either way. */
#define PTGPU_OPT_ONE_COUNTER 256u
/* The staging queue that closes the ragged end of a launch reserved through the eight counters holds 32 Ki rays instead of
 * 4 Mi, so that every such launch takes the piecewise path a real frame needs only when its streams end megarays apart: for
 * tests of that path. */
#define PTGPU_OPT_SMALL_STAGING 512u

/* Per-pixel primary-ray hit record, the `ray_triangle_hit_info` of
 * src/ray_program.h:29-56 (raycast parity dump). */
typedef struct ptgpu_hit {
    float t; /* +inf on miss */
    float A;
    float B;
    uint32_t object_index;
    uint32_t mesh_index;
    uint32_t triangle_index; /* index of the triangle's first entry in mesh indices */
    uint32_t hit;            /* 1 = forward hit (hit && t > 0) */
    uint32_t backface;
} ptgpu_hit;

/* Counters and timings of the last ptgpu_render_* call. */
typedef struct ptgpu_stats {
    uint64_t rays;           /* intersect_ray calls: camera + secondary + pass-through */
    uint64_t camera_samples; /* (pixel, sample) pairs generated */
    uint64_t extend_launches;
    uint64_t shade_launches;
    double extend_ms;  /* sum of HIP-event durations of the extend stage (instance pass + traversal) */
    double shade_ms;   /* same for the shade kernels */
    double generate_ms;
    double total_ms;   /* HIP-event time from first to last kernel of the call */
    uint64_t nodes_visited;    /* filled only by ptgpu_render_counted */
    uint64_t triangles_tested; /* idem */
    uint64_t instances_tested; /* idem */
    uint64_t instances_entered; /* idem */
    uint64_t wave_inner_iterations; /* idem: inner-node loop iterations summed over wavefronts */
    uint64_t wave_triangle_iterations; /* idem: triangle loop iterations summed over wavefronts */
    double traverse_ms; /* part of extend_ms spent in the traversal kernel (0 with PTGPU_EXTEND=simple) */
    uint64_t nodes_from_lds; /* ptgpu_render_counted: node visits served by the traversal kernel's LDS copy of the top nodes */
    uint32_t traverse_workgroups_per_cu; /* form of the traversal kernel the loaded scene runs with: 6, or 7 (four-wide nodes
                                            over a scene the caches hold); 0 with PTGPU_EXTEND=simple */
    uint32_t ragged_end_in_pieces; /* PBR launches whose output streams ended too far apart for the one-piece fix-up (see
                                      PTGPU_OPT_SMALL_STAGING): closed piece by piece, slower, same rays */
    uint64_t leaf_records_read; /* ptgpu_render_counted, scenes whose four-wide leaves are 64-byte records: records fetched (0 otherwise) */
    uint64_t rays_abandoned;    /* Chains of rays that were cut because they did not advance.  A pass-through ray (alpha != 1,
                                   src/ray_program.cc:265-271) starts at position - normal * 1e-5 with the interpolated
                                   normal; where that normal and the triangle's plane disagree about the side the ray came
                                   from (flipped vertex normals, smooth normals under strong non-uniform scale), the new origin
                                   is in front of the triangle again and the reference's loop (src/renderer.cc:111-117)
                                   never ends.  Here a chain of pass-through rays ends at the third visit in a row of one
                                   triangle, each within 0.001 of the last, and in any case at the eighth such step in a
                                   row; the frame is complete except for what those chains would still have added.  0 on
                                   every scene whose normals agree with its geometry. */
} ptgpu_stats;

typedef struct ptgpu_bvh_info {
    uint32_t n_nodes;  /* inner nodes of the flattened tree */
    uint32_t n_leaves;
    uint32_t n_triangles;
    uint32_t min_depth, max_depth;
    float mean_depth;
    uint32_t min_tris_in_leaf, max_tris_in_leaf;
    float mean_tris_in_leaf;
    uint64_t checksum; /* FNV-1a over leaf order of triangle indices and leaf sizes */
} ptgpu_bvh_info;

typedef struct ptgpu_ctx ptgpu_ctx;

/* Create a context on HIP device `device_id`.  Fails with PTGPU_E_NODEVICE if
 * there is no such device or it is not a gfx950 part. */
int ptgpu_create(int device_id, ptgpu_ctx** out);
void ptgpu_destroy(ptgpu_ctx* ctx);
const char* ptgpu_last_error(const ptgpu_ctx* ctx); /* ctx may be NULL: last create error */

/* Renderer::set_render_settings (src/renderer.cc:125). Takes effect on the
 * next ptgpu_load_scene for AS/program choices, like the reference, and
 * immediately for spp / bounces / seed / sample subset. */
int ptgpu_set_render_settings(ptgpu_ctx* ctx, const ptgpu_settings* settings);
int ptgpu_get_render_settings(const ptgpu_ctx* ctx, ptgpu_settings* out);

/* Takes effect at the next ptgpu_load_scene (build and kernel-form choices) or the next render (chunk, pools). */
int ptgpu_set_options(ptgpu_ctx* ctx, const ptgpu_options* options);
int ptgpu_get_options(const ptgpu_ctx* ctx, ptgpu_options* out);

/* Renderer::load_scene (src/renderer.cc:41-64): builds the two-level
 * acceleration structure on the host exactly like BVH_AS::BVH_AS
 * (src/acceleration_structure.cc:352-536), flattens it to the device layout
 * and uploads geometry, materials, textures and the environment map. */
int ptgpu_load_scene(ptgpu_ctx* ctx, const ptgpu_model* model, const ptgpu_envmap* env);

/* The scene of another context, copied device to device (over xGMI between two GPUs of a node): the reference builds
 * its acceleration structure once per Renderer::load_scene (src/renderer.cc:41-64), and a renderer that drives several
 * GPUs should too — ptgpu_load_scene on the first context, this call on the others.  `ctx` ends up as if
 * ptgpu_load_scene had been called on it with the source's model, environment and scene settings (acceleration
 * structure type, leaf size, program, wide-node flag, environment rotation are taken over; samples, bounces, seed,
 * sample / tile subset stay its own).  The source must not be rendering or loading during the call. */
int ptgpu_load_scene_from(ptgpu_ctx* ctx, const ptgpu_ctx* source);

/* Acceleration-structure builds (host or device side) this process has completed: one per ptgpu_load_scene,
 * none per ptgpu_load_scene_from. */
uint64_t ptgpu_host_build_count(void);

/* Renderer::update_camera_transform_state (src/renderer.cc:27-39).
 * aspect <= 0 selects the reference default 1.7777778, zfar <= 0 selects 1000. */
int ptgpu_update_camera_transform_state(ptgpu_ctx* ctx, const float position[3], const float direction[3],
                                        const float up[3], float yfov, float aspect, float znear, float zfar);
/* Same state set directly: NDC2World matrix and ray origin. */
int ptgpu_set_camera_raw(ptgpu_ctx* ctx, const float ndc2world[16], const float origin[3]);
int ptgpu_get_camera_raw(const ptgpu_ctx* ctx, float ndc2world[16], float origin[3]);

/* Renderer::get_scene_bound (src/renderer.cc:128-133). */
int ptgpu_get_scene_bound(const ptgpu_ctx* ctx, float bmin[3], float bmax[3]);

/* Renderer::render_frame (src/renderer.cc:77-123): renders width x height
 * pixels with the current settings and writes the mean colour, alpha = 1, as
 * width*height RGBA float rows (top row first) to host memory. */
int ptgpu_render_frame(ptgpu_ctx* ctx, int width, int height, float* out_rgba, ptgpu_stats* stats);

/* Device-resident variant for multi-GPU sample split: leaves the SUM over this
 * context's sample subset in `d_sum_rgba` (width*height*4 floats on the
 * context's device, alpha channel = number of samples summed) without dividing.
 * `stream` is a hipStream_t; the call returns after the work has completed on
 * that stream.  NULL does NOT mean the legacy default stream: it selects the
 * context's own non-blocking stream, which is ordered with nothing else — a
 * caller that chains other device work (a collective, a copy) around this call
 * must create a stream, pass its handle here and issue that work on it too. */
int ptgpu_render_frame_sums(ptgpu_ctx* ctx, int width, int height, float* d_sum_rgba, void* stream,
                            ptgpu_stats* stats);

/* Divide a summed RGBA buffer by its sample count and set alpha = 1
 * (the `get_mean()` of src/renderer.h:31 applied to all pixels), on device. */
int ptgpu_resolve_sums(ptgpu_ctx* ctx, int width, int height, float* d_sum_rgba, uint32_t total_samples,
                       void* stream);

/* Same frame as ptgpu_render_frame, traced by the counting variant of the
 * extend kernel: fills nodes_visited / triangles_tested / instances_* in stats. */
int ptgpu_render_counted(ptgpu_ctx* ctx, int width, int height, float* out_rgba, ptgpu_stats* stats);

/* Primary-ray hit records for sample index `sample` of every pixel
 * (what BVH_AS::intersect_ray returns for the camera rays, src/renderer.cc:110-115). */
int ptgpu_render_hits(ptgpu_ctx* ctx, int width, int height, uint32_t sample, ptgpu_hit* out);

/* General-purpose intersect_ray batch (IAccelerationStructure::intersect_ray,
 * src/acceleration_structure.h:66): n rays, origins/directions as xyz triples,
 * any_hit[i] != 0 selects any-hit traversal (NULL = closest hit). */
int ptgpu_intersect_rays(ptgpu_ctx* ctx, uint64_t n, const float* origins, const float* directions,
                         const uint8_t* any_hit, ptgpu_hit* out);

/* The same batch for callers that cannot allocate (tight serving loops,
 * harnesses that forbid hipMalloc); it has no counterpart in the reference.
 * (Code under HIP graph capture is the other caller of that kind, but a
 * captured call did not replay correctly when tried, so capture is NOT
 * supported: call it on a stream, outside capture.)  Inputs and the output are device memory on
 * the context's device (n origins and directions as xyz triples, n any-hit
 * bytes or NULL, n hit records out); everything else the traversal needs lives
 * in a work array that is part of the library's code object, so the call makes
 * no hipMalloc / hipFree and no other allocation.  Limits: at most
 * PTGPU_STATIC_MAX_RAYS rays per call (more: PTGPU_E_INVALID), and trees whose
 * traversal stack fits the array (deeper: PTGPU_E_NOMEM).  The work is queued
 * on `stream` (NULL = the context's own stream, see ptgpu_render_frame_sums);
 * wait != 0 returns after it has completed, wait == 0 right after queueing.
 * All calls on one GPU share that GPU's one array: calls that overlap in time
 * on the same GPU (two streams, two contexts) race and must be serialised by
 * the caller.  Results equal ptgpu_intersect_rays bit for bit. */
#define PTGPU_STATIC_MAX_RAYS (1u << 17)
int ptgpu_intersect_rays_static(ptgpu_ctx* ctx, uint32_t n, const float* d_origins, const float* d_directions,
                                const uint8_t* d_any_hit, ptgpu_hit* d_out, void* stream, int wait);

/* Tree statistics of one mesh BVH (MeshBVHData::collect_tree_info,
 * src/acceleration_structure.cc:538-570). */
int ptgpu_get_bvh_info(const ptgpu_ctx* ctx, uint32_t mesh_index, ptgpu_bvh_info* out);

/* Sub-pixel sample table (Renderer::generate_subsample_positions,
 * src/renderer.cc:135-155): writes spp (x, y) pairs. Host-only helper. */
int ptgpu_subsample_positions(uint32_t spp, float* out_xy);

#ifdef __cplusplus
}
#endif
#endif /* PTGPU_H */
