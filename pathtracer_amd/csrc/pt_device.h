// pt_device.h — device-side data layout and math of the MI355X path tracer.
//
// Layout (all arrays live in HBM, addressed through SceneDev):
//   nodes      64 B   binary BVH node holding BOTH children's boxes and references, so one inner-node
//                     visit is one 64-byte fetch (4 x dwordx4); half of a 128-byte L2 line.  The planes are
//                     grouped per axis (left lo, right lo, left hi, right hi) so that the slab arithmetic
//                     runs on register pairs.
//   tris       48 B   p1, p2 - p1, p3 - p1 + index of the triangle's first entry in the mesh index list +
//                     size of the leaf (valid in a leaf's first triangle) — 3 x dwordx4.
//   instances  64 B   world-space 14-DOP (7 min/max pairs) + object and mesh index.
//   xforms    128 B   inverse model and model matrices (3 rows used) + root reference + cull flag.
//   corners   144 B   per triangle (mesh order, not BVH order) its three vertices de-indexed:
//                     position normal tangent uv, 3 x dwordx4 each.
//   objects   256 B   model and normal matrices, first corner record of the mesh, material with its
//                     texture descriptors.
// Arithmetic follows the reference operation by operation (one binary32 rounding each; the build
// disables contraction), so hit records can be compared bit for bit with a CPU evaluation.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_LEAF_BIT 0x80000000u
#define PT_NO_NODE 0xffffffffu
#define PT_IDLE 0xfffffffeu      // traversal lane without a ray (never a valid reference)

namespace ptd {

struct alignas(16) DevNode {      // 64 B
    float l_lo[3], l_hi[3];       // left child box
    float r_lo[3], r_hi[3];       // right child box
    uint32_t left, right;         // child refs: inner -> node index, leaf -> PT_LEAF_BIT | first triangle
    uint32_t pad0, pad1;
};
struct alignas(16) DevTri {       // 48 B
    float p1[3], p2[3], p3[3];
    uint32_t index;               // triangleIndex of the reference: 3 * triangle number in the mesh
    uint32_t leaf_count;          // number of triangles of the leaf that starts here
    uint32_t pad;
};
// Traversal forms of the two records above, made once at scene load.
struct alignas(16) NodeQ {        // 64 B
    float x[4], y[4], z[4];       // per axis: left lo, right lo, left hi, right hi
    uint32_t left, right;
    uint32_t pad0, pad1;
};
// Four-wide node (optional form of the same trees): a binary node with inner children pulled up, its children's boxes
// quantised to 8 bits per plane inside the node's own box — 64 B for four children instead of 64 B for two.
// Plane value = origin[a] + scale[a] * q; a child's low planes are rounded down, its high planes up, so the stored box
// contains the exact one.  Byte k of qlo[a] / qhi[a] belongs to child k; unused slots have ref PT_NO_NODE.
struct alignas(64) Node4 {
    float origin[3];
    float scale[3];
    uint32_t qlo[3], qhi[3];
    uint32_t ref[4];
};
struct alignas(16) TriQ {         // 48 B
    float p1[3], e1[3], e2[3];    // e1 = p2 - p1, e2 = p3 - p1 (the first two operations of the triangle test)
    uint32_t index;
    uint32_t leaf_count;          // triangles of the leaf from this one on (1 = last of its leaf)
    uint32_t pad;
};
// One leaf of a four-wide tree as ONE aligned 64-byte record (large scenes, where the traversal is limited by the number of
// memory sectors it touches): the leaf's one or two triangles as their three or four distinct corner positions.  The first
// triangle is (v0, v1, v2) in the mesh's corner order; the second, when it shares an edge with the first, is three of
// v0..v3 picked by 2-bit selectors.  A leaf whose two triangles share no edge takes two records, the first flagged
// "continues".  The kernel forms p2 - p1 and p3 - p1 itself, the same two binary32 subtractions the triangle test of the
// reference starts with (acceleration_structure.cc:24-25), so the hit is the one the 48-byte records give, to the bit.
struct alignas(64) Leaf64 {
    float v[4][3];
    uint32_t index0, index1;      // triangleIndex of the first and the second triangle
    uint32_t info;                // bit 0: a second triangle; bit 1: the leaf continues in the next record;
                                  // bits 8-9, 10-11, 12-13: which of v0..v3 are the second triangle's p1, p2, p3
    uint32_t pad;
};
struct alignas(16) DevInstance {  // 64 B
    float dop[14];                // (min,max) for the 7 axes
    uint32_t object_index;
    uint32_t mesh_index;
};
// Node of the tree over the instances' boxes, stored in depth-first order and threaded: a ray that enters the box goes on
// to the next node (an inner node's first child follows it), a ray that does not goes to `skip` (the node after the
// subtree) — no stack.  A leaf (inst != PT_NO_NODE) names one instance.
struct alignas(16) TlasNode {     // 32 B
    float lo[3];
    uint32_t skip;
    float hi[3];
    uint32_t inst;
};
struct alignas(16) DevXform {     // 128 B
    float inv[12];                // inverse model matrix columns c0..c3, xyz rows
    float model[12];              // model matrix columns c0..c3, xyz rows
    uint32_t root;                // root reference of the mesh BVH (leaf bit set when the root is a leaf)
    uint32_t cull;                // 1 = back-face culling (mesh is not double sided)
    uint32_t object_index;
    uint32_t mesh_index;
    uint32_t root4;               // root reference into SceneDev::nodes4 (when the scene has four-wide nodes)
    uint32_t pad[3];
};
struct DevTexture {               // 12 B; width 0 = the material has no such texture
    int32_t width, height;
    uint32_t offset;              // first texel in SceneDev::texels
};
struct alignas(16) DevMaterial {  // shading form of ptgpu_material, its texture descriptors inline; 144 B
    float base_color[4];
    float emissive[3], emissive_strength;
    float attenuation[3], metallic;
    float roughness, ior, f0, transmission;
    float alpha_cutoff;
    uint32_t double_sided, has_volume, alpha_blending;
    DevTexture base_color_tex, mr_tex, normal_tex, transmission_tex, emissive_tex;
    uint32_t pad;
};
// Everything shading needs to know about an instance in ONE record (256 B), so that a hit costs three
// dependent fetches — hit record -> object -> triangle corners -> texels — instead of walking
// object -> mesh -> index list -> vertices -> material -> image table.
struct alignas(16) DevObject {
    float model[12];              // columns c0..c3 xyz
    float normal[12];             // NormalMatrix columns c0..c3 xyz
    uint32_t tri_base;            // first record of the object's mesh in SceneDev::corners
    uint32_t pad[3];
    DevMaterial mat;
};

struct SceneDev {
    const NodeQ* nodes;
    const TriQ* tris;
    const DevInstance* instances;
    const DevXform* xforms;
    const DevObject* objects;
    const float4* corners;        // per triangle in mesh order: its three vertices, 3 float4 each (144 B)
    const uint32_t* texels;       // RGBA8
    const float4* env;
    const float* srgb_lut;        // 256 entries: srgb_to_linear(i / 255)
    uint32_t n_instances;         // entries of `instances` / `xforms` (objects whose model matrix has no inverse are left out)
    uint32_t n_objects;           // entries of `objects` (every object of the model: an instance names its object by index)
    uint32_t n_top_nodes;         // nodes[0 .. n_top_nodes) are the most visited ones (kept in LDS by the traversal kernel)
    const TlasNode* tlas;         // threaded tree over the instances' boxes, or null: few instances, every ray tests them all
    uint32_t n_tlas;              // its node count
    const Node4* nodes4;          // four-wide form of the same trees, or null: the persistent kernel then walks `nodes`
    uint32_t n_top_nodes4;        // likewise for nodes4
    const Leaf64* leaves64;       // leaves of the four-wide trees as 64-byte records, or null: nodes4 refers to `tris`
    uint32_t seven_waves;         // the four-wide traversal kernel runs in its seven-workgroups-per-CU form (shorter LDS stack)
    uint32_t leaf_vote;           // weight of the lanes that hold a leaf when the traversal kernel chooses its next phase (1 or 4)
    int32_t env_w, env_h;
    float env_rot;
};

// ---------------------------------------------------------------------------------------------
// float3 helpers with the operation order glm uses
// ---------------------------------------------------------------------------------------------
struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 mk3(float s) { return f3{s, s, s}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return f3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ bool any_ne(f3 a, f3 b) { return a.x != b.x || a.y != b.y || a.z != b.z; }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross3(f3 a, f3 b) {
    return f3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
__device__ __forceinline__ f3 normalize3(f3 v) { return v * (1.0f / sqrtf(dot3(v, v))); }
__device__ __forceinline__ float length3(f3 v) { return sqrtf(dot3(v, v)); }
__device__ __forceinline__ float sel_min(float a, float b) { return (b < a) ? b : a; }  // glm::min
__device__ __forceinline__ float sel_max(float a, float b) { return (a < b) ? b : a; }  // glm::max
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return sel_min(sel_max(x, lo), hi); }
__device__ __forceinline__ float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
__device__ __forceinline__ f3 mix3(f3 a, f3 b, float t) { return f3{mixf(a.x, b.x, t), mixf(a.y, b.y, t), mixf(a.z, b.z, t)}; }
__device__ __forceinline__ float4 mix4(float4 a, float4 b, float t) {
    return make_float4(mixf(a.x, b.x, t), mixf(a.y, b.y, t), mixf(a.z, b.z, t), mixf(a.w, b.w, t));
}
// mat(3 rows of a column-major 4x4, 12 floats) * (v, w): (c0*x + c1*y) + (c2*z + c3*w)
__device__ __forceinline__ f3 xform(const float* m, f3 v, float w) {
    return f3{(m[0] * v.x + m[3] * v.y) + (m[6] * v.z + m[9] * w), (m[1] * v.x + m[4] * v.y) + (m[7] * v.z + m[10] * w),
              (m[2] * v.x + m[5] * v.y) + (m[8] * v.z + m[11] * w)};
}
// mat3 (columns t, b, n) * v: (c0*x + c1*y) + c2*z
__device__ __forceinline__ f3 mul_tbn(f3 t, f3 b, f3 n, f3 v) {
    return f3{t.x * v.x + b.x * v.y + n.x * v.z, t.y * v.x + b.y * v.y + n.y * v.z, t.z * v.x + b.z * v.y + n.z * v.z};
}

// ---------------------------------------------------------------------------------------------
// counter-based sampler: every ray carries a 32-bit key; uniforms and child keys are hashes of it
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t rng_camera_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
    uint32_t k = mix32(seed ^ 0x85ebca6bU);
    k = mix32(k + pixel);
    return mix32(k ^ (sample * 0x9e3779b9U + 0x7f4a7c15U));
}
__host__ __device__ __forceinline__ float rng_uniform(uint32_t key, uint32_t j) {
    return static_cast<float>(mix32(key + (j + 1u) * 0x9e3779b9U) >> 8) * (1.0f / 16777216.0f);
}
__host__ __device__ __forceinline__ uint32_t rng_child(uint32_t key, uint32_t lobe) {
    return mix32((key ^ 0x68bc21ebU) + (lobe + 1u) * 0x02e5be93U);
}
enum { LOBE_PASS = 0, LOBE_DIFFUSE = 1, LOBE_TRANSMIT = 2, LOBE_REFLECT = 3, LOBE_AO0 = 4 };

// ---------------------------------------------------------------------------------------------
// transcendental kernels: fixed polynomials evaluated with explicit fused multiply-adds, so the
// same formulas evaluated on a CPU give the same bits (std::sin / atan2 / asin / exp of the
// reference differ from these by a few ulp at most)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void pt_sincos(float x, float* s_out, float* c_out) {  // x in [0, 2*pi]
    float qf = rintf(x * 0.636619772367581343f);
    int q = static_cast<int>(qf);
    float r = fmaf(-qf, 1.57079625129699707031f, x);
    r = fmaf(-qf, 7.54978995489188216e-08f, r);
    float r2 = r * r;
    float sp = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = fmaf(sp, r2, -1.6666654611e-1f);
    float s = fmaf(sp * r2, r, r);
    float cp = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = fmaf(cp, r2, 4.166664568298827e-2f);
    float c = fmaf(cp * r2, r2, fmaf(-0.5f, r2, 1.0f));
    switch (q & 3) {
        case 0: *s_out = s; *c_out = c; break;
        case 1: *s_out = c; *c_out = -s; break;
        case 2: *s_out = -s; *c_out = -c; break;
        default: *s_out = -c; *c_out = s; break;
    }
}
__device__ __forceinline__ float pt_atan(float x) {
    float sign = 1.0f;
    if (x < 0.0f) { sign = -1.0f; x = -x; }
    float y;
    if (x > 2.414213562373095f) { y = 1.57079632679489661923f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.78539816339744830962f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    y = y + fmaf(p * z, x, x);
    return sign * y;
}
__device__ __forceinline__ float pt_atan2(float y, float x) {
    const float pi = 3.14159265358979323846264338327950288f;
    if (x == 0.0f) {
        if (y > 0.0f) return 1.57079632679489661923f;
        if (y < 0.0f) return -1.57079632679489661923f;
        return 0.0f;
    }
    float a = pt_atan(y / x);
    if (x < 0.0f) return (y < 0.0f) ? a - pi : a + pi;
    return a;
}
__device__ __forceinline__ float pt_asin(float x) {
    float sign = 1.0f, a = x;
    if (x < 0.0f) { sign = -1.0f; a = -x; }
    if (a > 1.0f) a = 1.0f;
    bool big = a > 0.5f;
    float z, w;
    if (big) { z = 0.5f * (1.0f - a); w = sqrtf(z); }
    else { w = a; z = w * w; }
    float p = fmaf(4.2163199048e-2f, z, 2.4181311049e-2f);
    p = fmaf(p, z, 4.5470025998e-2f);
    p = fmaf(p, z, 7.4953002686e-2f);
    p = fmaf(p, z, 1.6666752422e-1f);
    float r = fmaf(p * z, w, w);
    if (big) r = 1.57079632679489661923f - (r + r);
    return sign * r;
}
__device__ __forceinline__ float pt_exp(float x) {
    if (x != x) return x;
    if (x > 88.0f) return __builtin_inff();
    if (x < -87.0f) return 0.0f;
    float nf = floorf(fmaf(x, 1.44269504088896341f, 0.5f));
    int n = static_cast<int>(nf);
    float r = fmaf(nf, -0.693359375f, x);
    r = fmaf(nf, 2.12194440e-4f, r);
    float r2 = r * r;
    float p = fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float e = fmaf(p, r2, r) + 1.0f;
    return e * __uint_as_float(static_cast<uint32_t>(n + 127) << 23);
}

}  // namespace ptd
