/*
 * pt_oracle.h — C interface of the CPU parity oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a CPU restatement of the
 * reference renderer's hot path (zarond/PathTracer src/renderer.cc,
 * src/acceleration_structure.cc, src/ray_program.cc, src/brdf.cc and the
 * sampling half of src/cpu_framebuffer.h).  It exists so that tests, the smoke
 * check and the CPU-baseline leg of the benchmark have something to compare the
 * GPU library against; the shipped rendering path never links or calls it.
 *
 * PARITY UNPINNED: the reference has no golden vectors or known-answer tests
 * for this path and cannot be compiled here (all of its third-party
 * dependencies, glm included, are absent).  The only reference-held data this
 * oracle is checked against is the Fibonacci sequence of
 * tests/fibonacci_sequence.py.  Vector-math semantics follow the published
 * glm definitions (unknown pinned version).
 *
 * The struct layouts below are byte-identical to the ptgpu_* structs of the
 * GPU library's header so one set of buffers can feed both.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_vertex {
    float position[3];
    float normal[3];
    float tangent[4];
    float uv[2];
} orc_vertex;

typedef struct orc_mesh {
    uint32_t first_vertex, n_vertices, first_index, n_indices, material_index;
} orc_mesh;

typedef struct orc_object {
    float model_matrix[16];
    float normal_matrix[16];
    uint32_t mesh_index;
} orc_object;

typedef struct orc_material {
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
} orc_material;

typedef struct orc_image {
    int32_t width, height;
    uint64_t texel_offset;
} orc_image;

typedef struct orc_model {
    const orc_vertex* vertices;
    uint64_t n_vertices;
    const uint32_t* indices;
    uint64_t n_indices;
    const orc_mesh* meshes;
    uint32_t n_meshes;
    const orc_object* objects;
    uint32_t n_objects;
    const orc_material* materials;
    uint32_t n_materials;
    const orc_image* images;
    uint32_t n_images;
    const uint8_t* texels;
    uint64_t n_texels;
} orc_model;

typedef struct orc_envmap {
    int32_t width, height;
    const float* rgba;
} orc_envmap;

typedef struct orc_settings {
    uint32_t samples_per_pixel;
    uint32_t max_ray_bounces;
    uint32_t max_new_rays_per_bounce;
    uint32_t max_triangles_per_bvh_leaf;
    float envmap_rotation;
    int32_t program_mode;      /* 0 RayCaster, 1 AmbientOcclusion, 2 PBR */
    int32_t accel_struct_type; /* 0 Naive, 1 BVH */
    uint32_t seed;
    uint32_t sample_begin;
    uint32_t sample_step;
    uint32_t flags; /* ORC_FLAG_* */
    uint32_t tile_begin, tile_step; /* layout of the GPU library's settings; the oracle always renders every tile */
} orc_settings;
#define ORC_FLAG_SINGLE_PATH 1u /* PBR keeps one of the rays a hit would spawn (see Programs::pbr_single) */

typedef struct orc_hit {
    float t, A, B;
    uint32_t object_index, mesh_index, triangle_index, hit, backface;
} orc_hit;

typedef struct orc_stats {
    uint64_t rays;
    uint64_t camera_samples;
    uint64_t nodes_visited;     /* inner-node visits (each tests both children) */
    uint64_t triangles_tested;
    uint64_t instances_tested;  /* DOP tests */
    uint64_t instances_entered; /* mesh traversals started */
    double seconds;
    uint64_t chains_cut;        /* pass-through chains cut at the third visit in a row of one triangle or the eighth close step: the rule the GPU
                                   library counts in ptgpu_stats.rays_abandoned (the reference would not return) */
} orc_stats;

typedef struct orc_bvh_info {
    uint32_t n_nodes, n_leaves, n_triangles, min_depth, max_depth;
    float mean_depth;
    uint32_t min_tris_in_leaf, max_tris_in_leaf;
    float mean_tris_in_leaf;
    uint64_t checksum;
} orc_bvh_info;

/* RNG modes: 0 = counter-based per-ray keys (same definition as the GPU
 * library); 1 = std::minstd_rand + uniform_real_distribution<float> drawn in
 * depth-first order like the reference (seeded per image row from `seed`). */
enum { ORC_RNG_COUNTER = 0, ORC_RNG_MINSTD = 1 };
/* Transcendental math: 0 = libm (std::sin, std::atan2, ... as the reference
 * calls them); 1 = the portable polynomial kernels the GPU library uses. */
enum { ORC_MATH_LIBM = 0, ORC_MATH_PORTABLE = 1 };

typedef struct orc_scene orc_scene;

orc_scene* orc_scene_create(const orc_model* model, const orc_envmap* env, const orc_settings* settings);
void orc_scene_destroy(orc_scene* s);
void orc_set_settings(orc_scene* s, const orc_settings* settings); /* spp/bounces/seed/subset only */
void orc_set_rng_mode(orc_scene* s, int rng_mode);
void orc_set_math_mode(orc_scene* s, int math_mode);

void orc_update_camera_transform_state(orc_scene* s, const float position[3], const float direction[3],
                                       const float up[3], float yfov, float aspect, float znear, float zfar);
void orc_set_camera_raw(orc_scene* s, const float ndc2world[16], const float origin[3]);
void orc_get_camera_raw(const orc_scene* s, float ndc2world[16], float origin[3]);
void orc_get_scene_bound(const orc_scene* s, float bmin[3], float bmax[3]);

/* render_frame: mean colour (Welford, like the reference) when the sample
 * subset is the full set, alpha 1.  n_threads <= 0 uses hardware_concurrency. */
int orc_render_frame(orc_scene* s, int width, int height, float* out_rgba, int n_threads, orc_stats* stats);
/* Plain per-pixel SUM over this scene's sample subset, alpha = samples summed. */
int orc_render_frame_sums(orc_scene* s, int width, int height, float* out_rgba, int n_threads,
                          orc_stats* stats);
/* Restrict rendering to rows [y0, y1) (used for a bounded CPU-baseline sample). */
int orc_render_rows(orc_scene* s, int width, int height, int y0, int y1, float* out_rgba, int n_threads,
                    orc_stats* stats);
int orc_render_hits(orc_scene* s, int width, int height, uint32_t sample, orc_hit* out, int n_threads);
int orc_intersect_rays(orc_scene* s, uint64_t n, const float* origins, const float* directions,
                       const uint8_t* any_hit, orc_hit* out);
int orc_get_bvh_info(const orc_scene* s, uint32_t mesh_index, orc_bvh_info* out);
void orc_subsample_positions(uint32_t spp, float* out_xy);

/* ---- known-answer entry points for single functions --------------------- */
/* intersect_ray_triangle: out = {A, B, t, hit, backface} */
void orc_kat_ray_triangle(const float o[3], const float d[3], const float p1[3], const float p2[3],
                          const float p3[3], int backface_culling, float out[5]);
/* BBox::ray_volume_intersection(ray, inv_dir): out = {hit, t0, t1, forward_hit_distance} */
void orc_kat_bbox(const float bmin[3], const float bmax[3], const float o[3], const float d[3], float out[4]);
/* DOP over n points, then ray test with cached projections: out = {hit, t0, t1, forward_hit_distance} */
void orc_kat_dop(const float* points, int n, const float o[3], const float d[3], float out[4]);
/* CPUTexture<sdr_pixel>::sample_bilinear */
void orc_kat_sample_bilinear_rgba8(const uint8_t* texels, int w, int h, float u, float v, int srgb, float out[4]);
/* sample_environment */
void orc_kat_sample_environment(const float* rgba, int w, int h, const float dir[3], float rot, int math_mode,
                                float out[4]);
void orc_kat_cos_dir(float xi_x, float xi_y, int math_mode, float out[3]);
void orc_kat_ggx(float xi_x, float xi_y, float a, int math_mode, float out[3]);
void orc_kat_schlick(const float f0[3], const float f90[3], float c, float out[3]);
float orc_kat_v_smith(float NoV, float NoL, float a);
void orc_kat_refract(const float I[3], const float N[3], float eta, float out[3]);
void orc_kat_fibonacci2d(int i, float inv_n, float out[2]);
float orc_kat_srgb_to_linear(float c);
float orc_kat_linear_to_srgb(float c);
void orc_kat_float_pixel_to_srgb8(const float rgba[4], uint8_t out[4]);
uint32_t orc_kat_rng_camera_key(uint32_t seed, uint32_t pixel, uint32_t sample);
float orc_kat_rng_uniform(uint32_t key, uint32_t j);
uint32_t orc_kat_rng_child(uint32_t key, uint32_t lobe);

/* One shading call (IRayProgram::on_hit) on explicit inputs.  ray = {origin3,
 * direction3, payload3}, depth/any_hit/key as given; the hit record is given
 * explicitly.  Writes the returned radiance and up to max_out spawned rays
 * (each 9 floats + depth + any_hit + key); returns the number spawned. */
typedef struct orc_ray {
    float origin[3], direction[3], payload[3];
    uint32_t depth, any_hit, key;
} orc_ray;
int orc_kat_on_hit(orc_scene* s, const orc_ray* ray, const orc_hit* hit, float radiance[3], orc_ray* out_rays,
                   int max_out);

#ifdef __cplusplus
}
#endif
#endif
