// ptgpu_api.hip — implementation of the C ABI declared in include/ptgpu.h: context, scene upload and
// the host-side wavefront scheduler that drives generate -> extend -> shade until every ray queue drains.
//
// Scheduling.  The reference traces each camera sample depth first with a per-thread LIFO of rays
// (renderer.cc:100-117).  Here the LIFO holds ray *segments* of up to `chunk` rays: the newest segment
// is popped, extended and shaded; the rays it spawns become new segments on top.  Segments of
// alternating generations live in two bump-allocated pools, so the pool a segment came from is always
// freed from its top and live rays stay bounded by (fan-out x chunk) per generation however strongly
// the PBR estimator branches.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <queue>
#include <vector>

#include "ptgpu.h"
#include "pt_bvh.h"
#include "pt_kernels.h"
#include "pt_scene_build.h"

namespace {

thread_local std::string g_create_error;

// ---- work memory of ptgpu_intersect_rays_static ------------------------------------------------------
// That entry point may not allocate, so the buffers the traversal kernels need beyond its inputs and outputs — the ray
// records, the candidate lists, the hit queue, the claim cursors, the overflow list and the stack spill area — are
// carved out of one array that is part of the library's code object.  Every GPU the library runs on holds its own copy
// from the moment the code object is loaded there until the process ends.
constexpr uint32_t kStaticMaxRays = 1u << 17;     // rays per call
constexpr uint32_t kStaticBlocks = 256;           // resident traversal workgroups of that call (one spill column per thread)
constexpr uint32_t kStaticSpillDepth = 64;        // stack entries beyond the LDS part that the spill area holds per thread
constexpr size_t align256(size_t n) { return (n + 255) & ~static_cast<size_t>(255); }
struct StaticLayout {
    static constexpr size_t ray_o = 0;
    static constexpr size_t ray_d = ray_o + align256(16ull * kStaticMaxRays);
    static constexpr size_t ray_p = ray_d + align256(16ull * kStaticMaxRays);
    static constexpr size_t hit0 = ray_p + align256(16ull * kStaticMaxRays);
    static constexpr size_t hit1 = hit0 + align256(16ull * kStaticMaxRays);
    static constexpr size_t cand = hit1 + align256(4ull * kStaticMaxRays);
    static constexpr size_t overflow = cand + align256(8ull * ptd::kCandMax * kStaticMaxRays);
    static constexpr size_t cursors = overflow + align256(4ull * (kStaticMaxRays + 1));
    static constexpr size_t spill = cursors + align256(1024);
    static constexpr size_t total = spill + align256(4ull * kStaticSpillDepth * kStaticBlocks * ptd::kBlock);
};
static __device__ __align__(256) unsigned char scratch[StaticLayout::total];  // 34.6 MB
// In host code `scratch` is only the symbol's stand-in; the array's address on a device is asked for once per device
// (ptgpu_create, with that device current) and kept here.  Host code uses these pointers, never the name.
constexpr int kMaxDevices = 64;
unsigned char* g_scratch_ptr[kMaxDevices] = {};

// ---- host vector math with glm's operation order (camera state only) -----------------------------
struct H3 {
    float x, y, z;
};
inline H3 sub(H3 a, H3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline H3 add(H3 a, H3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float hdot(H3 a, H3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline H3 hcross(H3 a, H3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline H3 hnorm(H3 v) {
    float s = 1.0f / std::sqrt(hdot(v, v));
    return {v.x * s, v.y * s, v.z * s};
}
struct Mat4 {
    float m[16];  // column-major: m[c * 4 + r]
    float& at(int c, int r) { return m[c * 4 + r]; }
    float at(int c, int r) const { return m[c * 4 + r]; }
};
Mat4 mat_mul(const Mat4& a, const Mat4& b) {  // glm: ((a0*b0 + a1*b1) + a2*b2) + a3*b3 per column
    Mat4 r;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            float t = a.at(0, i) * b.at(j, 0);
            t = t + a.at(1, i) * b.at(j, 1);
            t = t + a.at(2, i) * b.at(j, 2);
            t = t + a.at(3, i) * b.at(j, 3);
            r.at(j, i) = t;
        }
    return r;
}
Mat4 mat_inverse(const Mat4& m) {  // glm::inverse(mat4): cofactors, then one reciprocal determinant
    auto e = [&](int c, int r) { return m.at(c, r); };
    const float c00 = e(2, 2) * e(3, 3) - e(3, 2) * e(2, 3), c02 = e(1, 2) * e(3, 3) - e(3, 2) * e(1, 3),
                c03 = e(1, 2) * e(2, 3) - e(2, 2) * e(1, 3);
    const float c04 = e(2, 1) * e(3, 3) - e(3, 1) * e(2, 3), c06 = e(1, 1) * e(3, 3) - e(3, 1) * e(1, 3),
                c07 = e(1, 1) * e(2, 3) - e(2, 1) * e(1, 3);
    const float c08 = e(2, 1) * e(3, 2) - e(3, 1) * e(2, 2), c10 = e(1, 1) * e(3, 2) - e(3, 1) * e(1, 2),
                c11 = e(1, 1) * e(2, 2) - e(2, 1) * e(1, 2);
    const float c12 = e(2, 0) * e(3, 3) - e(3, 0) * e(2, 3), c14 = e(1, 0) * e(3, 3) - e(3, 0) * e(1, 3),
                c15 = e(1, 0) * e(2, 3) - e(2, 0) * e(1, 3);
    const float c16 = e(2, 0) * e(3, 2) - e(3, 0) * e(2, 2), c18 = e(1, 0) * e(3, 2) - e(3, 0) * e(1, 2),
                c19 = e(1, 0) * e(2, 2) - e(2, 0) * e(1, 2);
    const float c20 = e(2, 0) * e(3, 1) - e(3, 0) * e(2, 1), c22 = e(1, 0) * e(3, 1) - e(3, 0) * e(1, 1),
                c23 = e(1, 0) * e(2, 1) - e(2, 0) * e(1, 1);
    const float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11},
                f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
    const float v0[4] = {e(1, 0), e(0, 0), e(0, 0), e(0, 0)}, v1[4] = {e(1, 1), e(0, 1), e(0, 1), e(0, 1)},
                v2[4] = {e(1, 2), e(0, 2), e(0, 2), e(0, 2)}, v3[4] = {e(1, 3), e(0, 3), e(0, 3), e(0, 3)};
    const float sa[4] = {+1, -1, +1, -1}, sb[4] = {-1, +1, -1, +1};
    Mat4 inv;
    for (int i = 0; i < 4; ++i) {
        inv.at(0, i) = ((v1[i] * f0[i] - v2[i] * f1[i]) + v3[i] * f2[i]) * sa[i];
        inv.at(1, i) = ((v0[i] * f0[i] - v2[i] * f3[i]) + v3[i] * f4[i]) * sb[i];
        inv.at(2, i) = ((v0[i] * f1[i] - v1[i] * f3[i]) + v3[i] * f5[i]) * sa[i];
        inv.at(3, i) = ((v0[i] * f2[i] - v1[i] * f4[i]) + v2[i] * f5[i]) * sb[i];
    }
    const float d0 = e(0, 0) * inv.at(0, 0), d1 = e(0, 1) * inv.at(1, 0), d2 = e(0, 2) * inv.at(2, 0),
                d3 = e(0, 3) * inv.at(3, 0);
    const float ood = 1.0f / ((d0 + d1) + (d2 + d3));
    for (float& v : inv.m) v = v * ood;
    return inv;
}

// ---- device memory ---------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    hipError_t ensure(size_t n) {
        if (n <= bytes) return hipSuccess;
        release();  // first: the pools are tens of GB, old and new together may not fit
        if (alloc_fails_now()) return hipErrorOutOfMemory;
        hipError_t e = hipMalloc(&p, n ? n : 16);
        if (e == hipSuccess) bytes = n;
        else p = nullptr;
        return e;
    }
    // Test hook: PTGPU_TEST_FAIL_ALLOC=k makes the k-th allocation from then on (counted from 1) fail as if the device
    // were out of memory, so the error paths can be exercised without exhausting a 288 GB device.
    // (counted from the moment the variable's text last changed, so one process can arm it repeatedly: "3:a", "3:b", ...)
    static bool alloc_fails_now() {
        const char* v = std::getenv("PTGPU_TEST_FAIL_ALLOC");
        if (!v || !*v) return false;
        static std::string armed;
        static long count = 0;
        if (armed != v) {
            armed = v;
            count = 0;
        }
        return ++count == std::atol(v);
    }
    template <typename T>
    T* as() const { return static_cast<T*>(p); }
};

struct Pool {
    DevBuf o, d, p;
    uint64_t cap = 0, top = 0;
    ptd::RayQueue at(uint64_t off) const { return {o.as<float4>() + off, d.as<float4>() + off, p.as<float4>() + off}; }
};

}  // namespace

struct ptgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;
    ptgpu_settings settings{};        // current
    ptgpu_settings scene_settings{};  // as of the last load_scene (AS type, leaf size, program, env rotation)
    bool scene_loaded = false, camera_set = false;
    ptgpu_options options{};          // tuning (ptgpu_set_options); zero fields mean "the library's choice"
    bool has_nodes4 = false;          // the loaded scene carries four-wide nodes
    bool has_leaves64 = false;        // ... whose leaves are 64-byte records

    // scene
    DevBuf nodes, nodes4, leaves64, tlas, tris, instances, xforms, objects, corners, texels, env, lut;
    ptd::SceneDev scene{};
    std::vector<ptgpu_bvh_info> bvh_info;
    std::vector<uint32_t> object_mesh;  // object -> mesh index (for hit dumps)
    DevBuf d_object_mesh;               // the same table on the device (part of the scene, filled by load_scene)
    unsigned char* static_scratch = nullptr;  // this device's copy of the library's built-in work array
    uint32_t max_depth = 0;
    float scene_min[3] = {0, 0, 0}, scene_max[3] = {0, 0, 0};

    // camera (Renderer::NDC2WorldMatrix_, origin_)
    float ndc2world[16];
    float origin[3];

    // work memory
    // Upper bound of rays per launch and of camera samples per batch (PTGPU_CHUNK).  Large on purpose: every launch
    // of the persistent extend kernel ends with a tail in which a few long rays keep nearly empty wavefronts
    // alive, so throughput grows with the launch size (about +18 % from 8 Mi to 64 Mi rays), and the memory is there.
    uint32_t chunk = 1u << 26;
    uint32_t batch_cap = 0;        // camera samples per batch the ray pools are sized for
    uint32_t launch_cap = 0;       // rays per launch the work buffers are sized for (also the candidate list stride)
    Pool pools[2];
    DevBuf hit0, hit1, spill, fb, subs, counter, trace_counters, out_rgba, ray_cursor, cand, overflow;
    DevBuf stream_cursors, staging_o, staging_d, staging_p;  // the PBR kernel's multi-counter reservation (launch_shade)
    static constexpr uint32_t kStagingRays = 1u << 22;       // rays the fix-up of a launch's ragged end can hold
    bool persistent = true;        // extend kernel form (PTGPU_EXTEND=simple selects one ray per lane)
    uint32_t persist_blocks = 0;   // resident workgroups of the persistent extend kernel
    uint32_t* host_counter = nullptr;  // pinned
    hipEvent_t ev[9] = {};         // (7, 8: around the camera-ray kernel of a batch whose ray count needs no readback)
    std::vector<float> subs_host;  // the subsample table on the device is that of subs_spp samples per pixel (0: none yet)
    uint32_t subs_spp = 0;

    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        error = buf;
        return code;
    }
};

namespace {

#define PT_HIP(ctx, call)                                                                               \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return (ctx)->fail(e_ == hipErrorOutOfMemory ? PTGPU_E_NOMEM : PTGPU_E_HIP, "%s failed: %s", #call, \
                               hipGetErrorString(e_));                                                  \
    } while (0)
// Wait for the stream, then ask whether every kernel launched since the last check was accepted by the runtime: a
// rejected launch does not fail the synchronise, it just never wrote what the host is about to read.
#define PT_SYNC(ctx, st)                                                                                \
    do {                                                                                                \
        PT_HIP(ctx, hipStreamSynchronize(st));                                                          \
        const char* k_ = "";                                                                            \
        const hipError_t le_ = ptd::take_launch_error(&k_);                                             \
        if (le_ != hipSuccess) return (ctx)->fail(PTGPU_E_HIP, "launch of %s failed: %s", k_, hipGetErrorString(le_)); \
    } while (0)

template <typename T>
int upload(ptgpu_ctx* c, DevBuf& buf, const T* src, size_t n) {
    PT_HIP(c, buf.ensure(std::max<size_t>(n, 1) * sizeof(T)));
    if (n) PT_HIP(c, hipMemcpy(buf.p, src, n * sizeof(T), hipMemcpyHostToDevice));
    return PTGPU_OK;
}

// A new context's options start from PTGPU_* environment variables: a convenience for the A/B tools (tools/ab_*.sh),
// which alternate settings on one machine without touching the caller.  This is the one place the library reads them;
// everything else reads ctx->options, and ptgpu_set_options replaces what was found here.
ptgpu_options options_from_environment() {
    ptgpu_options o{};
    auto number = [](const char* name, long lo, long hi) -> uint32_t {
        const char* v = std::getenv(name);
        if (!v || !*v) return 0u;
        const long k = std::atol(v);
        return (k >= lo && k <= hi) ? static_cast<uint32_t>(k) : 0u;
    };
    auto is = [](const char* name, const char* value) {
        const char* v = std::getenv(name);
        return v && std::strcmp(v, value) == 0;
    };
    o.chunk_rays = number("PTGPU_CHUNK", 1024, 1l << 28);
    o.pool_gib = number("PTGPU_POOL_GIB", 1, 256);
    o.traverse_waves = number("PTGPU_TRAVERSE_WAVES", 6, 7);
    o.leaf_vote = number("PTGPU_LEAF_VOTE", 1, 16);
    o.build_threads = number("PTGPU_BUILD_THREADS", 1, 64);
    o.intersect_chunk = number("PTGPU_INTERSECT_CHUNK", 1, 1l << 30);
    o.persist_blocks = number("PTGPU_PERSIST_BLOCKS", 1, 1 << 20);
    if (is("PTGPU_EXTEND", "simple")) o.flags |= PTGPU_OPT_EXTEND_SIMPLE;
    if (is("PTGPU_WIDE_SPLITS", "reference")) o.flags |= PTGPU_OPT_REFERENCE_SPLITS;
    if (is("PTGPU_WIDE_COLLAPSE", "greedy")) o.flags |= PTGPU_OPT_GREEDY_COLLAPSE;
    if (is("PTGPU_TLAS", "linear")) o.flags |= PTGPU_OPT_LINEAR_TLAS;
    if (is("PTGPU_RESERVE", "one-counter")) o.flags |= PTGPU_OPT_ONE_COUNTER;
    if (is("PTGPU_LEAF_RECORDS", "1")) o.flags |= PTGPU_OPT_LEAF_RECORDS;
    if (is("PTGPU_LEAF_RECORDS", "0")) o.flags |= PTGPU_OPT_NO_LEAF_RECORDS;
    const uint32_t verbose = number("PTGPU_VERBOSE", 1, 9);
    if (verbose >= 1) o.flags |= PTGPU_OPT_VERBOSE;
    if (verbose >= 2) o.flags |= PTGPU_OPT_VERBOSE_LAUNCHES;
    return o;
}
void apply_options(ptgpu_ctx* c) {
    const bool persistent = !(c->options.flags & PTGPU_OPT_EXTEND_SIMPLE);
    if (persistent != c->persistent) c->scene_loaded = false;  // the loaded scene was laid out for the other kernel
    c->persistent = persistent;
    c->chunk = c->options.chunk_rays >= 1024 ? (std::min<uint32_t>(c->options.chunk_rays, 1u << 28) & ~63u) : (1u << 26);
    c->persist_blocks = 0;
}

std::vector<float> subsample_table(uint32_t spp) {  // Renderer::generate_subsample_positions, renderer.cc:135-155
    std::vector<float> t(2 * static_cast<size_t>(spp));
    const int root = static_cast<int>(sqrtf(static_cast<float>(spp)));
    if (static_cast<uint32_t>(root * root) == spp) {
        for (uint32_t i = 0; i < spp; ++i) {
            t[2 * i] = (static_cast<float>(i / root) + 0.5f) / static_cast<float>(root);
            t[2 * i + 1] = (static_cast<float>(i % root) + 0.5f) / static_cast<float>(root);
        }
    } else {
        const float inv = 1.0f / static_cast<float>(spp);
        for (uint32_t i = 0; i < spp; ++i) {  // fibonacci2D, brdf.cc:73-78
            const float f = static_cast<float>(i);
            t[2 * i] = (f + 0.5f) * inv;
            t[2 * i + 1] = std::fmod((f + 1.0f) * 1.618034f, 1.0f);
        }
    }
    return t;
}

struct FrameGeometry {
    ptd::CameraDev cam;
    uint32_t n_tiles;
    uint32_t fb_stride;
};
FrameGeometry frame_geometry(const ptgpu_ctx* c, int w, int h) {
    FrameGeometry g;
    std::memcpy(g.cam.ndc2world, c->ndc2world, 64);
    std::memcpy(g.cam.origin, c->origin, 12);
    g.cam.inv_w = 1.0f / static_cast<float>(w);  // renderer.cc:89-90
    g.cam.inv_h = 1.0f / static_cast<float>(h);
    g.cam.width = w;
    g.cam.height = h;
    g.cam.tiles_x = (w + ptd::kTile - 1) / ptd::kTile;
    g.cam.tiles_y = (h + ptd::kTile - 1) / ptd::kTile;
    g.n_tiles = static_cast<uint32_t>(g.cam.tiles_x) * g.cam.tiles_y;
    g.fb_stride = g.n_tiles * 64u;
    return g;
}

uint32_t fanout_of(const ptgpu_settings& s) {
    if (s.program_mode == PTGPU_PROGRAM_PBR)  // pass-through + diffuse + transmission + reflection, or the one of them kept
        return (s.flags & PTGPU_FLAG_SINGLE_PATH) ? 1 : 4;
    if (s.program_mode == PTGPU_PROGRAM_AO) return s.max_new_rays_per_bounce;
    return 0;
}

// K = camera samples per batch (sizes the ray pools), L = largest launch (sizes hit records, candidate lists, overflow list)
// Stack entries beyond the LDS part of the traversal kernel in use (the seven-wave kernel keeps fewer in LDS).
uint32_t spill_depth_of(const ptgpu_ctx* c) {
    const uint32_t in_lds = c->scene.seven_waves ? ptd::kStackLds7 : ptd::kStackLds;
    return c->max_depth > in_lds ? c->max_depth - in_lds : 0;
}

int ensure_work_memory(ptgpu_ctx* c, uint32_t fanout, uint32_t bounces, uint64_t K, uint64_t L) {
    (void)ptd::take_launch_error(nullptr);  // a launch failure left over from a call that ended early is not this call's
    K = std::max<uint64_t>(K, c->batch_cap);
    L = std::max<uint64_t>(std::max<uint64_t>(L, K), c->launch_cap);
    // live rays per generation are bounded by fanout * chunk; generations alternate between the pools
    uint64_t gens = std::min<uint64_t>(bounces, 254) / 2 + 2;
    // (+ the slots a launch of the PBR kernel may leave unused at the end of its output before its ragged end is closed)
    uint64_t cap = K + static_cast<uint64_t>(fanout) * K * gens + (fanout ? ptd::shade_stream_slack() : 0u);
    // at most 32 GiB per pool (ptgpu_options.pool_gib overrides); the scheduler splits segments beyond that
    const uint64_t pool_gib = c->options.pool_gib ? c->options.pool_gib : 32;
    const uint64_t cap_limit = (pool_gib << 30) / 48;
    cap = std::min(cap, std::max<uint64_t>(cap_limit, K + static_cast<uint64_t>(fanout) * 64 + (fanout ? ptd::shade_stream_slack() : 0u)));
    // A failed allocation leaves its buffer empty (ensure frees first).  The sizes recorded in the context must never
    // claim more than the buffers hold, so they are cleared before anything is touched and set again once all of it is there.
    c->batch_cap = 0;
    c->launch_cap = 0;
    for (Pool& p : c->pools) {
        if (p.cap < cap) {
            p.cap = 0;
            PT_HIP(c, p.o.ensure(cap * 16));
            PT_HIP(c, p.d.ensure(cap * 16));
            PT_HIP(c, p.p.ensure(cap * 16));
            p.cap = cap;
        }
    }
    PT_HIP(c, c->hit0.ensure(L * 16));
    PT_HIP(c, c->hit1.ensure(L * 4));
    const uint32_t spill_depth = spill_depth_of(c);
    if (c->persistent && !c->persist_blocks) {
        c->persist_blocks = ptd::extend_persistent_grid(c->device, c->scene.seven_waves != 0);
        if (c->options.persist_blocks) c->persist_blocks = c->options.persist_blocks;
        if (c->options.flags & PTGPU_OPT_VERBOSE) std::fprintf(stderr, "[ptgpu] persistent extend grid: %u workgroups\n", c->persist_blocks);
    }
    const uint64_t spill_columns = c->persistent ? static_cast<uint64_t>(c->persist_blocks) * ptd::kBlock : L;
    PT_HIP(c, c->spill.ensure(static_cast<size_t>(spill_depth) * spill_columns * 4 + 16));
    PT_HIP(c, c->ray_cursor.ensure(1024));
    if (c->persistent) PT_HIP(c, c->cand.ensure(static_cast<size_t>(ptd::kCandMax) * L * sizeof(uint2)));
    if (c->persistent) PT_HIP(c, c->overflow.ensure((L + 1) * sizeof(uint32_t)));
    PT_HIP(c, c->counter.ensure(16));
    if (fanout && !(c->options.flags & PTGPU_OPT_ONE_COUNTER)) {
        PT_HIP(c, c->stream_cursors.ensure(ptd::kShadeStreamWords * sizeof(uint32_t)));
        PT_HIP(c, c->staging_o.ensure(static_cast<size_t>(ptgpu_ctx::kStagingRays) * 16));
        PT_HIP(c, c->staging_d.ensure(static_cast<size_t>(ptgpu_ctx::kStagingRays) * 16));
        PT_HIP(c, c->staging_p.ensure(static_cast<size_t>(ptgpu_ctx::kStagingRays) * 16));
    }
    PT_HIP(c, c->trace_counters.ensure(sizeof(ptd::TraceCounters)));
    c->batch_cap = static_cast<uint32_t>(K);
    c->launch_cap = static_cast<uint32_t>(L);
    return PTGPU_OK;
}

void run_extend(ptgpu_ctx* c, hipStream_t st, const ptd::RayQueue& rq, uint32_t n, const ptd::HitQueue& hits,
                ptd::TraceCounters* tc, hipEvent_t between = nullptr) {
    const uint32_t spill_depth = spill_depth_of(c);
    if (c->persistent)
        ptd::launch_extend_persistent(st, c->scene, rq, n, hits, ptd::CandQueue{c->cand.as<uint2>(), c->launch_cap},
                                      c->spill.as<uint32_t>(), spill_depth, c->persist_blocks, c->ray_cursor.as<uint32_t>(),
                                      c->overflow.as<uint32_t>(), tc,
                                      between);
    else
        ptd::launch_extend(st, c->scene, rq, n, hits, c->spill.as<uint32_t>(), spill_depth, c->launch_cap, tc);
}

// The subsample positions of `spp` samples per pixel in c->subs; computed and copied only when spp changes (a preview loop
// renders the same settings frame after frame, and the copy from pageable memory costs more than its 8 bytes per sample).
int upload_subsamples(ptgpu_ctx* c, uint32_t spp, hipStream_t st) {
    if (c->subs_spp == spp && c->subs.p) return PTGPU_OK;
    c->subs_spp = 0;
    c->subs_host = subsample_table(spp);  // (kept in the context: the copy may still be reading it when this returns)
    PT_HIP(c, c->subs.ensure(c->subs_host.size() * 4));
    PT_HIP(c, hipMemcpyAsync(c->subs.p, c->subs_host.data(), c->subs_host.size() * 4, hipMemcpyHostToDevice, st));
    PT_HIP(c, hipStreamSynchronize(st));  // later frames may run on another stream
    c->subs_spp = spp;
    return PTGPU_OK;
}

struct RenderOut {
    float* d_rgba;    // device RGBA row-major destination (may be null)
    bool mean;        // divide by the number of samples rendered
};

// The frame loop: Renderer::render_frame (renderer.cc:77-123) as a wavefront schedule.
int render_core(ptgpu_ctx* c, int width, int height, hipStream_t st, bool counted, ptgpu_stats* stats, float* d_rgba,
                bool mean) {
    if (!c->scene_loaded) return c->fail(PTGPU_E_STATE, "render_frame: no scene loaded (a component is null)");
    if (!c->camera_set) return c->fail(PTGPU_E_STATE, "render_frame: camera transform state not set");
    if (width <= 0 || height <= 0) return c->fail(PTGPU_E_INVALID, "render_frame: bad framebuffer size %dx%d", width, height);
    PT_HIP(c, hipSetDevice(c->device));
    const ptgpu_settings s = c->settings;
    const int program = c->scene_settings.program_mode;
    const uint32_t spp = s.samples_per_pixel ? s.samples_per_pixel : 1;
    const uint32_t step = s.sample_step ? s.sample_step : 1;
    const uint32_t begin = s.sample_begin;
    const uint32_t n_slots = begin < spp ? (spp - begin + step - 1) / step : 0;
    ptgpu_settings fan_src = s;
    fan_src.program_mode = program;
    fan_src.max_new_rays_per_bounce = c->scene_settings.max_new_rays_per_bounce;
    const uint32_t fanout = fanout_of(fan_src);
    const uint32_t ao_samples = c->scene_settings.max_new_rays_per_bounce;
    const uint32_t bounces = s.max_ray_bounces & 0xffu;  // static_cast<uint8_t>, renderer.cc:73

    const FrameGeometry g = frame_geometry(c, width, height);
    // Camera samples are generated in equal batches of at most `chunk`; a batch is also the largest launch.  A batch
    // is a range of whole sample slots of every tile whenever one slot of the frame fits a chunk (tile-major queue
    // order, see generate_kernel); frames larger than that fall back to ranges of the linear sample-major index.
    // tile split (multi-GPU alternative to the sample split): only tiles tile_begin, tile_begin + tile_step, ... are rendered
    const uint32_t tile_step = s.tile_step ? s.tile_step : 1;
    const ptd::TileSubset tiles{s.tile_begin, tile_step,
                                s.tile_begin < g.n_tiles ? (g.n_tiles - s.tile_begin + tile_step - 1) / tile_step : 0u};
    const uint64_t total = static_cast<uint64_t>(n_slots) * tiles.count * 64ull;
    const uint64_t per_slot = static_cast<uint64_t>(tiles.count) * 64ull;
    uint32_t slots_per_batch = 0;
    uint64_t K;
#ifndef PT_SAMPLE_MAJOR
#define PT_SAMPLE_MAJOR 0  // 1: developer builds, the linear sample-major order for every frame (measured slower, see EXPERIMENTS)
#endif
    if (per_slot <= c->chunk && n_slots && !PT_SAMPLE_MAJOR) {
        const uint64_t fit = c->chunk / per_slot;
        const uint64_t n_batches = (n_slots + fit - 1) / fit;
        slots_per_batch = static_cast<uint32_t>((n_slots + n_batches - 1) / n_batches);
        K = slots_per_batch * per_slot;
    } else {
        const uint64_t n_batches = std::max<uint64_t>((total + c->chunk - 1) / c->chunk, 1);
        K = std::max<uint64_t>((((total + n_batches - 1) / n_batches) + 63ull) & ~63ull, 64);
    }
    // Spawned generations outgrow the camera batch (up to ~3x for the branching PBR estimator); launching a whole
    // generation at once saves the tail of every extra launch, so launches may be four batches long.
    const uint64_t L_want = fanout > 1 ? std::min<uint64_t>(4 * K, 1ull << 28) : (fanout == 1 ? 2 * K : K);
    int rc = ensure_work_memory(c, fanout, bounces, K, L_want);
    if (rc) return rc;
    const uint64_t L = c->launch_cap;
    PT_HIP(c, c->fb.ensure(static_cast<size_t>(g.fb_stride) * 3 * 4 * ptd::kFrameBuckets));
    ptd::FrameDev fb{c->fb.as<float>(), g.fb_stride, c->counter.as<uint32_t>() + 2};  // (word 0 of that buffer is the queue counter)
    PT_HIP(c, hipMemsetAsync(fb.abandoned, 0, 4, st));
    ptd::launch_clear(st, fb.sum, static_cast<uint64_t>(g.fb_stride) * 3 * ptd::kFrameBuckets);  // framebuffer.clear, renderer.cc:85
    if (const int rc_subs = upload_subsamples(c, spp, st)) return rc_subs;
    ptd::TraceCounters* d_tc = counted ? c->trace_counters.as<ptd::TraceCounters>() : nullptr;
    if (counted) PT_HIP(c, hipMemsetAsync(d_tc, 0, sizeof(ptd::TraceCounters), st));

    ptd::SceneDev sc = c->scene;
    ptd::HitQueue hits{c->hit0.as<float4>(), c->hit1.as<uint32_t>()};
    uint32_t* d_count = c->counter.as<uint32_t>();
    ptd::ShadeStreams streams;
    const bool use_streams = fanout && program == PTGPU_PROGRAM_PBR && !(c->options.flags & PTGPU_OPT_ONE_COUNTER);
    if (use_streams) {
        streams.cursors = c->stream_cursors.as<uint32_t>();
        streams.staging = ptd::RayQueue{c->staging_o.as<float4>(), c->staging_d.as<float4>(), c->staging_p.as<float4>()};
        streams.staging_capacity = (c->options.flags & PTGPU_OPT_SMALL_STAGING) ? (1u << 15) : ptgpu_ctx::kStagingRays;
    }
    const uint64_t stream_slack = use_streams ? ptd::shade_stream_slack() : 0u;

    ptgpu_stats out{};
    PT_HIP(c, hipEventRecord(c->ev[4], st));
    // Generation-ordered schedule.  A frame says "drain pool `pool` down to floor_in, spawned rays go on top of
    // the other pool (which holds older rays up to floor_out)".  A generation is consumed from the top of its pool
    // in launches of up to `chunk` rays; when it is gone the spawned generation is drained the same way, pools
    // swapped.  If the other pool cannot take a launch's worst-case output, the rays spawned so far are drained
    // first (depth first), so live rays stay bounded whatever the branching factor is.
    struct Frame {
        int pool;
        uint64_t floor_in, floor_out;
    };
    constexpr uint64_t kMinLaunch = 1u << 16;
#ifdef PT_DEV_DUMP_RAYS  // developer builds only (tools/sort_experiment.sh): one launch's rays written to a file
    const char* dump_path = std::getenv("PTGPU_DUMP_RAYS");
    const uint64_t dump_launch = std::getenv("PTGPU_DUMP_LAUNCH") ? std::strtoull(std::getenv("PTGPU_DUMP_LAUNCH"), nullptr, 10) : 0;
#endif
    std::vector<Frame> frames;
    float ms = 0.0f;
    const bool verbose_launches = (c->options.flags & PTGPU_OPT_VERBOSE_LAUNCHES) != 0;
    // Rays carry their own depth, so generations may mix: when what a generation spawned still fits a launch together
    // with the next batch's camera rays, it is not launched on its own — the camera rays go on top of it in its pool.
    // Launches stay full that way (a generation shrinks from bounce to bounce; the alpha pass-through chains at the
    // end of a batch are slivers of a few thousand rays), and every launch saved is a tail saved.
    const uint64_t carry_below = (L > K) ? L - K + 1 : K / 32;
    int gen_pool = 0;  // pool that receives the next batch's camera rays (the one holding carried rays, if any)
    bool generate_untimed = false;  // events 7 / 8 hold a camera-ray kernel's time that has not been read yet
    // A frame that is ONE launch with nothing to read back (a preview frame: one batch, no spawned rays) is not waited for
    // launch by launch either: its times are read after the frame's last kernel.
    const bool one_launch = !fanout && total <= K && total <= L;
    bool launch_untimed = false;
    c->pools[0].top = c->pools[1].top = 0;
    for (uint64_t first = 0; first < total; first += K) {
        const uint32_t n_gen = static_cast<uint32_t>(std::min<uint64_t>(K, total - first));
        const bool last_batch = first + K >= total;
        Pool& gp = c->pools[gen_pool];
        if (gp.top + n_gen > gp.cap) return c->fail(PTGPU_E_NOMEM, "ray pool too small for a batch of camera rays");
        // A frame of whole tiles has a ray for every index of the batch: its count is known here, and the host does not
        // wait for the kernel (the kernel's time is read at the next point where it waits anyway).
        const bool count_known = ptd::generate_is_dense(g.cam);
        PT_HIP(c, hipMemsetAsync(d_count, 0, 4, st));
        PT_HIP(c, hipEventRecord(c->ev[count_known ? 7 : 0], st));
        if (slots_per_batch) {
            const uint32_t slot0 = static_cast<uint32_t>(first / per_slot);
            ptd::launch_generate(st, g.cam, c->subs.as<float2>(), begin, step, bounces, s.seed, slot0, n_gen,
                                 static_cast<uint32_t>(n_gen / per_slot), tiles, gp.at(gp.top), d_count);
        } else {
            ptd::launch_generate(st, g.cam, c->subs.as<float2>(), begin, step, bounces, s.seed, first, n_gen, 0, tiles, gp.at(gp.top),
                                 d_count);
        }
        PT_HIP(c, hipEventRecord(c->ev[count_known ? 8 : 1], st));
        uint32_t n_cam = n_gen;
        if (count_known) {
            generate_untimed = true;
        } else {
            PT_HIP(c, hipMemcpyAsync(c->host_counter, d_count, 4, hipMemcpyDeviceToHost, st));
            PT_SYNC(c, st);
            PT_HIP(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
            out.generate_ms += ms;
            n_cam = *c->host_counter;
        }
        out.camera_samples += n_cam;
        gp.top += n_cam;
        if (!gp.top) continue;
        frames.assign(1, Frame{gen_pool, 0, 0});

        while (!frames.empty()) {
            const Frame f = frames.back();
            Pool& in = c->pools[f.pool];
            Pool& op = c->pools[f.pool ^ 1];
            if (in.top <= f.floor_in) {  // generation consumed: drain what it spawned
                frames.pop_back();
                if (op.top > f.floor_out) {
                    if (!last_batch && frames.empty() && op.top < carry_below) {
                        gen_pool = f.pool ^ 1;  // carried into the next batch (only ever the whole remainder of this one)
                        break;
                    }
                    frames.push_back({f.pool ^ 1, f.floor_out, f.floor_in});
                }
                continue;
            }
            uint64_t n64 = std::min<uint64_t>(L, in.top - f.floor_in);
            if (fanout) {
                // (the PBR kernel's output may end up to stream_slack slots past its rays before the ragged end is closed)
                const uint64_t free_slots = op.cap - op.top;
                const uint64_t room = (free_slots > stream_slack ? free_slots - stream_slack : 0) / fanout;
                if (room < std::min<uint64_t>(n64, kMinLaunch)) {
                    if (op.top > f.floor_out) {
                        frames.push_back({f.pool ^ 1, f.floor_out, in.top});  // depth first: drain the spawned rays, then resume
                        continue;
                    }
                    if (room == 0)  // nothing to drain and no room for one ray's offspring
                        return c->fail(PTGPU_E_NOMEM, "ray pool exhausted (%llu rays live); lower the chunk size",
                                       static_cast<unsigned long long>(in.top + op.top));
                }
                n64 = std::min(n64, room);
            }
            const uint32_t n = static_cast<uint32_t>(n64);
            const uint64_t offset = in.top - n;
            if (verbose_launches)
                std::fprintf(stderr, "[ptgpu] launch %llu: %u rays from pool %d (live %llu + %llu, depth of schedule %zu)\n",
                             static_cast<unsigned long long>(out.extend_launches), n, f.pool, static_cast<unsigned long long>(in.top),
                             static_cast<unsigned long long>(op.top), frames.size());
            const ptd::RayQueue rq = in.at(offset);
#ifdef PT_DEV_DUMP_RAYS
            if (dump_path && out.extend_launches == dump_launch) {  // one launch's rays as raw float4 o[n], d[n]
                std::vector<float4> buf(2ull * n);
                PT_HIP(c, hipMemcpy(buf.data(), rq.o, n * 16ull, hipMemcpyDeviceToHost));
                PT_HIP(c, hipMemcpy(buf.data() + n, rq.d, n * 16ull, hipMemcpyDeviceToHost));
                if (FILE* f = std::fopen(dump_path, "wb")) {
                    std::fwrite(buf.data(), 16, buf.size(), f);
                    std::fclose(f);
                }
            }
#endif
            PT_HIP(c, hipEventRecord(c->ev[0], st));
            run_extend(c, st, rq, n, hits, d_tc, c->ev[6]);
            PT_HIP(c, hipEventRecord(c->ev[1], st));
            if (fanout) PT_HIP(c, hipMemsetAsync(d_count, 0, 4, st));
            PT_HIP(c, hipEventRecord(c->ev[2], st));
            ptd::launch_shade(st, program, sc, rq, n, hits, op.at(op.top), d_count, fb, ao_samples,
                              (s.flags & PTGPU_FLAG_SINGLE_PATH) != 0, use_streams ? &streams : nullptr);
            PT_HIP(c, hipEventRecord(c->ev[3], st));
            if (fanout) PT_HIP(c, hipMemcpyAsync(c->host_counter, d_count, 4, hipMemcpyDeviceToHost, st));
            const bool through_streams = ptd::shade_reserves_through_streams(program, n, use_streams ? &streams : nullptr);
            if (through_streams)
                PT_HIP(c, hipMemcpyAsync(c->host_counter + 1, streams.cursors + ptd::kShadeStreamWords - 1, 4, hipMemcpyDeviceToHost, st));
            if (one_launch) {
                launch_untimed = true;
                out.rays += n;
                ++out.extend_launches;
                ++out.shade_launches;
                in.top = offset;
                continue;
            }
            PT_SYNC(c, st);
            if (generate_untimed) {
                PT_HIP(c, hipEventElapsedTime(&ms, c->ev[7], c->ev[8]));
                out.generate_ms += ms;
                generate_untimed = false;
            }
            if (through_streams && c->host_counter[1] != 0) {
                // the streams ended further apart than the staging queue holds (a launch whose workgroups' yields follow
                // their position modulo the number of streams): slower, piece by piece
                PT_HIP(c, ptd::close_ragged_end_piecewise(st, op.at(op.top), d_count, streams, c->host_counter));
                ++out.ragged_end_in_pieces;
            }
            PT_HIP(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
            out.extend_ms += ms;
            if (c->persistent) {
                PT_HIP(c, hipEventElapsedTime(&ms, c->ev[6], c->ev[1]));
                out.traverse_ms += ms;
            }
            PT_HIP(c, hipEventElapsedTime(&ms, c->ev[2], c->ev[3]));
            out.shade_ms += ms;
            out.rays += n;
            ++out.extend_launches;
            ++out.shade_launches;
            in.top = offset;  // the launched rays were the newest of their pool
            if (fanout) op.top += *c->host_counter;
        }
    }
    if (d_rgba) {
        const float n_samples = static_cast<float>(n_slots);
        ptd::launch_resolve(st, fb, g.cam, d_rgba, mean ? n_samples : 1.0f, mean ? 1.0f : n_samples);
    }
    PT_HIP(c, hipEventRecord(c->ev[5], st));
    PT_HIP(c, hipMemcpyAsync(c->host_counter + 2, fb.abandoned, 4, hipMemcpyDeviceToHost, st));
    PT_SYNC(c, st);
    if (generate_untimed) {
        PT_HIP(c, hipEventElapsedTime(&ms, c->ev[7], c->ev[8]));
        out.generate_ms += ms;
    }
    if (launch_untimed) {
        PT_HIP(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
        out.extend_ms += ms;
        if (c->persistent) {
            PT_HIP(c, hipEventElapsedTime(&ms, c->ev[6], c->ev[1]));
            out.traverse_ms += ms;
        }
        PT_HIP(c, hipEventElapsedTime(&ms, c->ev[2], c->ev[3]));
        out.shade_ms += ms;
    }
    PT_HIP(c, hipEventElapsedTime(&ms, c->ev[4], c->ev[5]));
    out.total_ms = ms;
    out.rays_abandoned = c->host_counter[2];
    out.traverse_workgroups_per_cu = c->persistent ? (c->scene.nodes4 && c->scene.seven_waves ? 7u : 6u) : 0u;
    if (counted) {
        ptd::TraceCounters tc;
        PT_HIP(c, hipMemcpy(&tc, d_tc, sizeof(tc), hipMemcpyDeviceToHost));
        out.nodes_visited = tc.nodes;
        out.triangles_tested = tc.tris;
        out.instances_tested = tc.inst_tested;
        out.instances_entered = tc.inst_entered;
        out.wave_inner_iterations = tc.wave_inner;
        out.wave_triangle_iterations = tc.wave_tri;
        out.nodes_from_lds = tc.census[1][0];
        out.leaf_records_read = tc.leaf_records;
        if (c->options.flags & PTGPU_OPT_VERBOSE) {
            std::fprintf(stderr, "[ptgpu] node visits served from LDS: %llu of %llu; rays that touch no instance: %llu\n", tc.census[1][0], tc.nodes,
                         tc.census[1][1]);
            static const char* names[4] = {"service", "-", "inner", "leaf"};
            for (int ph = 0; ph < 4; ++ph) {
                const double steps = static_cast<double>(std::max<unsigned long long>(tc.census[ph][1], 1));
                std::fprintf(stderr, "[ptgpu] census %-6s executions %llu steps %llu mean lanes: idle %.1f next %.1f node %.1f leaf %.1f\n",
                             names[ph], tc.census[ph][0], tc.census[ph][1], tc.census[ph][2] / steps, tc.census[ph][3] / steps,
                             tc.census[ph][4] / steps, tc.census[ph][5] / steps);
            }
        }
    }
    if (stats) *stats = out;
    return PTGPU_OK;
}

void decode_hits(const ptgpu_ctx* c, const std::vector<float4>& h0, const std::vector<uint32_t>& h1, size_t n, ptgpu_hit* out) {
    for (size_t i = 0; i < n; ++i) {
        ptgpu_hit& h = out[i];
        const bool hit = (h1[i] & 0x80000000u) != 0;
        h.hit = hit ? 1u : 0u;
        h.t = hit ? h0[i].x : INFINITY;
        h.A = hit ? h0[i].y : 0.0f;
        h.B = hit ? h0[i].z : 0.0f;
        uint32_t tri;
        std::memcpy(&tri, &h0[i].w, 4);
        h.triangle_index = hit ? tri : 0u;
        h.object_index = hit ? (h1[i] & 0x3fffffffu) : 0u;
        h.mesh_index = hit ? c->object_mesh[h.object_index] : 0u;
        h.backface = (hit && (h1[i] & 0x40000000u)) ? 1u : 0u;
    }
}

}  // namespace

extern "C" {

int ptgpu_create(int device_id, ptgpu_ctx** out) {
    if (!out) return PTGPU_E_INVALID;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_error = std::string("no HIP device: ") + hipGetErrorString(e);
        return PTGPU_E_NODEVICE;
    }
    if (device_id < 0 || device_id >= n) {
        g_create_error = "device index out of range";
        return PTGPU_E_NODEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) {
        g_create_error = "hipGetDeviceProperties failed";
        return PTGPU_E_NODEVICE;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only";
        return PTGPU_E_NODEVICE;
    }
    // owned through ptgpu_destroy from here on, so that a failure below releases the stream, events and pinned memory made so far
    std::unique_ptr<ptgpu_ctx, void (*)(ptgpu_ctx*)> ctx(new ptgpu_ctx, &ptgpu_destroy);
    ctx->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&ctx->host_counter), 64) != hipSuccess) {
        g_create_error = "stream / pinned memory creation failed";
        return PTGPU_E_HIP;
    }
    for (hipEvent_t& ev : ctx->ev)
        if (hipEventCreate(&ev) != hipSuccess) {
            ev = nullptr;
            g_create_error = "hipEventCreate failed";
            return PTGPU_E_HIP;
        }
    if (device_id < kMaxDevices) {
        if (!g_scratch_ptr[device_id]) {
            void* p = nullptr;
            const hipError_t es = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
            if (es != hipSuccess || !p) {
                g_create_error = std::string("hipGetSymbolAddress of the built-in work array failed: ") + hipGetErrorString(es);
                return PTGPU_E_HIP;
            }
            g_scratch_ptr[device_id] = static_cast<unsigned char*>(p);
        }
        ctx->static_scratch = g_scratch_ptr[device_id];
    }
    ptgpu_settings d{};
    d.samples_per_pixel = 1;  // defaults of arguments.cc:36-101
    d.max_ray_bounces = 4;
    d.max_new_rays_per_bounce = 0;
    d.max_triangles_per_bvh_leaf = 8;
    d.program_mode = PTGPU_PROGRAM_PBR;
    d.accel_struct_type = PTGPU_AS_BVH;
    d.sample_step = 1;
    d.tile_step = 1;
    ctx->settings = d;
    ctx->options = options_from_environment();
    apply_options(ctx.get());
    *out = ctx.release();
    return PTGPU_OK;
}

void ptgpu_destroy(ptgpu_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (hipEvent_t& ev : c->ev)
        if (ev) (void)hipEventDestroy(ev);
    if (c->host_counter) (void)hipHostFree(c->host_counter);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* ptgpu_last_error(const ptgpu_ctx* c) { return c ? c->error.c_str() : g_create_error.c_str(); }

int ptgpu_set_render_settings(ptgpu_ctx* c, const ptgpu_settings* s) {
    if (!c || !s) return PTGPU_E_INVALID;
    if (s->program_mode < 0 || s->program_mode > 2 || s->accel_struct_type < 0 || s->accel_struct_type > 2)
        return c->fail(PTGPU_E_INVALID, "set_render_settings: program_mode / accel_struct_type out of range");
    if (s->flags & ~(PTGPU_FLAG_SINGLE_PATH | PTGPU_FLAG_WIDE_NODES)) return c->fail(PTGPU_E_INVALID, "set_render_settings: unknown flag bits 0x%x", s->flags);
    c->settings = *s;
    return PTGPU_OK;
}
int ptgpu_set_options(ptgpu_ctx* c, const ptgpu_options* o) {
    if (!c || !o) return PTGPU_E_INVALID;
    if (o->flags & ~(PTGPU_OPT_EXTEND_SIMPLE | PTGPU_OPT_REFERENCE_SPLITS | PTGPU_OPT_GREEDY_COLLAPSE | PTGPU_OPT_LINEAR_TLAS |
                     PTGPU_OPT_VERBOSE | PTGPU_OPT_VERBOSE_LAUNCHES | PTGPU_OPT_LEAF_RECORDS | PTGPU_OPT_NO_LEAF_RECORDS | PTGPU_OPT_ONE_COUNTER | PTGPU_OPT_SMALL_STAGING))
        return c->fail(PTGPU_E_INVALID, "set_options: unknown flag bits 0x%x", o->flags);
    if ((o->chunk_rays && (o->chunk_rays < 1024 || o->chunk_rays > (1u << 28))) || o->pool_gib > 256 ||
        (o->traverse_waves && o->traverse_waves != 6 && o->traverse_waves != 7) || o->leaf_vote > 16 || o->build_threads > 64)
        return c->fail(PTGPU_E_INVALID, "set_options: a value is outside its range (see ptgpu.h)");
    c->options = *o;
    apply_options(c);
    return PTGPU_OK;
}
int ptgpu_get_options(const ptgpu_ctx* c, ptgpu_options* out) {
    if (!c || !out) return PTGPU_E_INVALID;
    *out = c->options;
    return PTGPU_OK;
}
int ptgpu_get_render_settings(const ptgpu_ctx* c, ptgpu_settings* out) {
    if (!c || !out) return PTGPU_E_INVALID;
    *out = c->settings;
    return PTGPU_OK;
}

// SceneDev's pointers = this context's scene buffers (absent optional arrays: null).
static void point_scene_at_buffers(ptgpu_ctx* c) {
    ptd::SceneDev& sc = c->scene;
    sc.nodes = c->nodes.as<ptd::NodeQ>();
    sc.tris = c->tris.as<ptd::TriQ>();
    sc.instances = c->instances.as<ptd::DevInstance>();
    sc.xforms = c->xforms.as<ptd::DevXform>();
    sc.objects = c->objects.as<ptd::DevObject>();
    sc.corners = c->corners.as<float4>();
    sc.texels = c->texels.as<uint32_t>();
    sc.env = c->env.as<float4>();
    sc.srgb_lut = c->lut.as<float>();
    sc.nodes4 = c->has_nodes4 ? c->nodes4.as<ptd::Node4>() : nullptr;
    sc.leaves64 = (c->has_nodes4 && c->has_leaves64) ? c->leaves64.as<ptd::Leaf64>() : nullptr;
    sc.tlas = sc.n_tlas ? c->tlas.as<ptd::TlasNode>() : nullptr;
}

// Device copies of a built scene and the SceneDev that points at them.
static int upload_built_scene(ptgpu_ctx* c, const pth::BuiltScene& b, const ptgpu_model* model, const ptgpu_envmap* env) {
    const float white[4] = {1.0f, 1.0f, 1.0f, 1.0f};  // CPUTexture<hdr_pixel>::create_white_texture, cpu_framebuffer.cc:104
    const float* env_px = white;
    int env_w = 1, env_h = 1;
    if (env && env->rgba) {
        if (env->width <= 0 || env->height <= 0) return c->fail(PTGPU_E_INVALID, "environment map has a bad size");
        env_px = env->rgba;
        env_w = env->width;
        env_h = env->height;
    }
    int rc;
    if ((rc = upload(c, c->nodes, b.nodes.data(), b.nodes.size()))) return rc;
    if ((rc = upload(c, c->tris, b.tris.data(), b.tris.size()))) return rc;
    if (!b.nodes4.empty() && (rc = upload(c, c->nodes4, b.nodes4.data(), b.nodes4.size()))) return rc;
    if (!b.leaves64.empty() && (rc = upload(c, c->leaves64, b.leaves64.data(), b.leaves64.size()))) return rc;
    if (!b.tlas.empty() && (rc = upload(c, c->tlas, b.tlas.data(), b.tlas.size()))) return rc;
    if ((rc = upload(c, c->instances, b.instances.data(), b.instances.size()))) return rc;
    if ((rc = upload(c, c->xforms, b.xforms.data(), b.xforms.size()))) return rc;
    if ((rc = upload(c, c->objects, b.objects.data(), b.objects.size()))) return rc;
    if ((rc = upload(c, c->corners, b.corners.data(), b.corners.size()))) return rc;
    if ((rc = upload(c, c->texels, reinterpret_cast<const uint32_t*>(model->texels), static_cast<size_t>(model->n_texels)))) return rc;
    if ((rc = upload(c, c->env, env_px, static_cast<size_t>(env_w) * env_h * 4))) return rc;
    if ((rc = upload(c, c->lut, b.srgb_lut, 256))) return rc;
    if ((rc = upload(c, c->d_object_mesh, b.object_mesh.data(), b.object_mesh.size()))) return rc;

    ptd::SceneDev& sc = c->scene;
    sc.n_instances = static_cast<uint32_t>(b.instances.size());
    sc.n_objects = static_cast<uint32_t>(b.objects.size());
    sc.n_top_nodes = b.n_top;
    sc.n_tlas = static_cast<uint32_t>(b.tlas.size());
    sc.n_top_nodes4 = b.n_top4;
    sc.seven_waves = b.seven_waves ? 1u : 0u;
    sc.leaf_vote = b.leaf_vote;
    sc.env_w = env_w;
    sc.env_h = env_h;
    c->has_nodes4 = !b.nodes4.empty();
    c->has_leaves64 = !b.leaves64.empty();
    point_scene_at_buffers(c);
    c->bvh_info = b.bvh_info;
    c->object_mesh = b.object_mesh;
    c->max_depth = b.max_depth;
    std::memcpy(c->scene_min, b.scene_min, 12);
    std::memcpy(c->scene_max, b.scene_max, 12);
    return PTGPU_OK;
}

int ptgpu_load_scene(ptgpu_ctx* c, const ptgpu_model* model, const ptgpu_envmap* env) {
    if (!c || !model) return PTGPU_E_INVALID;
    PT_HIP(c, hipSetDevice(c->device));
    c->scene_loaded = false;
    std::string err;
    if (!pth::validate_model(*model, &err)) return c->fail(PTGPU_E_INVALID, "%s", err.c_str());  // what the kernels index with
    const ptgpu_settings s = c->settings;
    pth::BuildOptions bo;
    bo.max_leaf = s.max_triangles_per_bvh_leaf;
    bo.naive = s.accel_struct_type == PTGPU_AS_NAIVE;
    bo.want_wide = (s.flags & PTGPU_FLAG_WIDE_NODES) && c->persistent;
    bo.reference_splits = (c->options.flags & PTGPU_OPT_REFERENCE_SPLITS) != 0;
    bo.greedy_collapse = (c->options.flags & PTGPU_OPT_GREEDY_COLLAPSE) != 0;
    bo.linear_tlas = (c->options.flags & PTGPU_OPT_LINEAR_TLAS) != 0;
    bo.traverse_waves = c->options.traverse_waves;
    bo.leaf_vote = c->options.leaf_vote;
    bo.leaf_records = (c->options.flags & PTGPU_OPT_LEAF_RECORDS) ? 1 : ((c->options.flags & PTGPU_OPT_NO_LEAF_RECORDS) ? -1 : 0);
    bo.build_threads = c->options.build_threads;
    bo.program_mode = s.program_mode;
    bo.verbose = (c->options.flags & PTGPU_OPT_VERBOSE) != 0;
    if (s.accel_struct_type == PTGPU_AS_LBVH) {
        hipStream_t st = c->stream;
        bo.mesh_builder = [st](const ptgpu_vertex* v, const uint32_t* idx, uint32_t n_idx, uint32_t max_leaf, pth::MeshTree& tree, std::string* e) {
            return pth::build_mesh_tree_lbvh(v, idx, n_idx, max_leaf, st, tree, e);
        };
    }
    pth::BuiltScene built;
    int code = PTGPU_E_INVALID;
    if (!pth::build_scene(*model, bo, built, &err, &code)) return c->fail(code, "%s", err.c_str());
    const int rc = upload_built_scene(c, built, model, env);
    if (rc) return rc;
    c->scene.env_rot = s.envmap_rotation;
    c->persist_blocks = 0;  // the resident grid depends on the kernel form: sized again at the next render
    c->scene_settings = s;
    c->scene_loaded = true;
    return PTGPU_OK;
}

int ptgpu_load_scene_from(ptgpu_ctx* c, const ptgpu_ctx* src) {
    if (!c || !src) return PTGPU_E_INVALID;
    if (c == src) return c->fail(PTGPU_E_INVALID, "load_scene_from: source and destination are the same context");
    if (!src->scene_loaded) return c->fail(PTGPU_E_STATE, "load_scene_from: the source context has no scene loaded");
    if (c->persistent != src->persistent) return c->fail(PTGPU_E_INVALID, "load_scene_from: the contexts run different extend kernels");
    PT_HIP(c, hipSetDevice(c->device));
    c->scene_loaded = false;
    // device to device, buffer by buffer: over xGMI between two GPUs of a node, inside HBM on one GPU
    auto copy = [&](DevBuf& dst, const DevBuf& from) -> int {
        if (!from.p || !from.bytes) return PTGPU_OK;
        PT_HIP(c, dst.ensure(from.bytes));
        if (c->device == src->device) PT_HIP(c, hipMemcpy(dst.p, from.p, from.bytes, hipMemcpyDeviceToDevice));
        else PT_HIP(c, hipMemcpyPeer(dst.p, c->device, from.p, src->device, from.bytes));
        return PTGPU_OK;
    };
    ptgpu_ctx* d = c;
    const std::pair<DevBuf*, const DevBuf*> pairs[] = {
        {&d->nodes, &src->nodes}, {&d->tris, &src->tris}, {&d->nodes4, &src->nodes4}, {&d->leaves64, &src->leaves64}, {&d->tlas, &src->tlas},
        {&d->instances, &src->instances}, {&d->xforms, &src->xforms}, {&d->objects, &src->objects}, {&d->corners, &src->corners},
        {&d->texels, &src->texels}, {&d->env, &src->env}, {&d->lut, &src->lut}, {&d->d_object_mesh, &src->d_object_mesh}};
    for (const auto& pr : pairs) {
        const int rc = copy(*pr.first, *pr.second);
        if (rc) return rc;
    }
    c->scene = src->scene;  // the scalars; the pointers are set to this context's buffers next
    c->has_nodes4 = src->has_nodes4;
    c->has_leaves64 = src->has_leaves64;
    point_scene_at_buffers(c);
    c->bvh_info = src->bvh_info;
    c->object_mesh = src->object_mesh;
    c->max_depth = src->max_depth;
    std::memcpy(c->scene_min, src->scene_min, 12);
    std::memcpy(c->scene_max, src->scene_max, 12);
    c->persist_blocks = 0;
    // what load_scene reads from the settings is what the SOURCE was loaded with; the rest of this context's settings stand
    c->scene_settings = src->scene_settings;
    c->settings.accel_struct_type = src->scene_settings.accel_struct_type;
    c->settings.program_mode = src->scene_settings.program_mode;
    c->settings.max_triangles_per_bvh_leaf = src->scene_settings.max_triangles_per_bvh_leaf;
    c->settings.max_new_rays_per_bounce = src->scene_settings.max_new_rays_per_bounce;
    c->settings.envmap_rotation = src->scene_settings.envmap_rotation;
    c->settings.flags = (c->settings.flags & ~PTGPU_FLAG_WIDE_NODES) | (src->scene_settings.flags & PTGPU_FLAG_WIDE_NODES);
    c->scene_loaded = true;
    return PTGPU_OK;
}

uint64_t ptgpu_host_build_count(void) { return pth::host_build_count(); }

int ptgpu_update_camera_transform_state(ptgpu_ctx* c, const float position[3], const float direction[3], const float up[3],
                                        float yfov, float aspect, float znear, float zfar) {
    if (!c || !position || !direction || !up) return PTGPU_E_INVALID;
    const H3 pos{position[0], position[1], position[2]}, dir{direction[0], direction[1], direction[2]}, upv{up[0], up[1], up[2]};
    // glm::lookAt (right-handed) with the translation column cleared (renderer.cc:30, 36-37)
    const H3 f = hnorm(sub(add(pos, dir), pos));
    const H3 s = hnorm(hcross(f, upv));
    const H3 u = hcross(s, f);
    Mat4 view{};
    view.at(0, 0) = s.x; view.at(1, 0) = s.y; view.at(2, 0) = s.z;
    view.at(0, 1) = u.x; view.at(1, 1) = u.y; view.at(2, 1) = u.z;
    view.at(0, 2) = -f.x; view.at(1, 2) = -f.y; view.at(2, 2) = -f.z;
    view.at(3, 3) = 1.0f;
    // glm::perspectiveRH, clip depth -1..1 (renderer.cc:31-35)
    const float asp = aspect > 0.0f ? aspect : 1.77777777777777777f;
    const float zf = zfar > 0.0f ? zfar : 1000.f;
    const float th = std::tan(yfov / 2.0f);
    Mat4 proj{};
    proj.at(0, 0) = 1.0f / (asp * th);
    proj.at(1, 1) = 1.0f / th;
    proj.at(2, 2) = -(zf + znear) / (zf - znear);
    proj.at(2, 3) = -1.0f;
    proj.at(3, 2) = -(2.0f * zf * znear) / (zf - znear);
    const Mat4 inv = mat_inverse(mat_mul(proj, view));  // renderer.cc:38
    std::memcpy(c->ndc2world, inv.m, 64);
    std::memcpy(c->origin, position, 12);
    c->camera_set = true;
    return PTGPU_OK;
}
int ptgpu_set_camera_raw(ptgpu_ctx* c, const float ndc2world[16], const float origin[3]) {
    if (!c || !ndc2world || !origin) return PTGPU_E_INVALID;
    std::memcpy(c->ndc2world, ndc2world, 64);
    std::memcpy(c->origin, origin, 12);
    c->camera_set = true;
    return PTGPU_OK;
}
int ptgpu_get_camera_raw(const ptgpu_ctx* c, float ndc2world[16], float origin[3]) {
    if (!c || !c->camera_set) return PTGPU_E_STATE;
    std::memcpy(ndc2world, c->ndc2world, 64);
    std::memcpy(origin, c->origin, 12);
    return PTGPU_OK;
}
int ptgpu_get_scene_bound(const ptgpu_ctx* c, float bmin[3], float bmax[3]) {
    if (!c || !bmin || !bmax) return PTGPU_E_INVALID;
    if (!c->scene_loaded) {  // Renderer::get_scene_bound without an AS returns an empty BBox (renderer.cc:128-133)
        for (int i = 0; i < 3; ++i) {
            bmin[i] = 3.402823466e+38f;
            bmax[i] = -3.402823466e+38f;
        }
        return PTGPU_OK;
    }
    std::memcpy(bmin, c->scene_min, 12);
    std::memcpy(bmax, c->scene_max, 12);
    return PTGPU_OK;
}

static int render_to_host(ptgpu_ctx* c, int width, int height, float* out_rgba, ptgpu_stats* stats, bool counted) {
    if (!c || !out_rgba) return PTGPU_E_INVALID;
    if (width <= 0 || height <= 0) return c->fail(PTGPU_E_INVALID, "render_frame: bad framebuffer size %dx%d", width, height);
    PT_HIP(c, hipSetDevice(c->device));
    const size_t bytes = static_cast<size_t>(width) * height * 16;
    PT_HIP(c, c->out_rgba.ensure(bytes));
    int rc = render_core(c, width, height, c->stream, counted, stats, c->out_rgba.as<float>(), true);
    if (rc) return rc;
    PT_HIP(c, hipMemcpy(out_rgba, c->out_rgba.p, bytes, hipMemcpyDeviceToHost));
    return PTGPU_OK;
}
int ptgpu_render_frame(ptgpu_ctx* c, int width, int height, float* out_rgba, ptgpu_stats* stats) {
    return render_to_host(c, width, height, out_rgba, stats, false);
}
int ptgpu_render_counted(ptgpu_ctx* c, int width, int height, float* out_rgba, ptgpu_stats* stats) {
    return render_to_host(c, width, height, out_rgba, stats, true);
}
int ptgpu_render_frame_sums(ptgpu_ctx* c, int width, int height, float* d_sum_rgba, void* stream, ptgpu_stats* stats) {
    if (!c || !d_sum_rgba) return PTGPU_E_INVALID;
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : c->stream;
    return render_core(c, width, height, st, false, stats, d_sum_rgba, false);
}
int ptgpu_resolve_sums(ptgpu_ctx* c, int width, int height, float* d_sum_rgba, uint32_t total_samples, void* stream) {
    if (!c || !d_sum_rgba || width <= 0 || height <= 0 || total_samples == 0) return PTGPU_E_INVALID;
    PT_HIP(c, hipSetDevice(c->device));
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : c->stream;
    ptd::launch_divide_rgba(st, d_sum_rgba, static_cast<uint64_t>(width) * height, static_cast<float>(total_samples));
    PT_SYNC(c, st);
    return PTGPU_OK;
}

int ptgpu_render_hits(ptgpu_ctx* c, int width, int height, uint32_t sample, ptgpu_hit* out) {
    if (!c || !out) return PTGPU_E_INVALID;
    if (!c->scene_loaded || !c->camera_set) return c->fail(PTGPU_E_STATE, "render_hits: scene or camera not set");
    if (width <= 0 || height <= 0) return c->fail(PTGPU_E_INVALID, "render_hits: bad size");
    const uint32_t spp = c->settings.samples_per_pixel ? c->settings.samples_per_pixel : 1;
    if (sample >= spp) return c->fail(PTGPU_E_INVALID, "render_hits: sample %u >= samples per pixel %u", sample, spp);
    PT_HIP(c, hipSetDevice(c->device));
    const FrameGeometry g = frame_geometry(c, width, height);
    const uint64_t total = static_cast<uint64_t>(g.n_tiles) * 64ull;
    const uint64_t K = std::min<uint64_t>(std::min<uint64_t>(c->chunk, 1u << 22), total);
    int rc = ensure_work_memory(c, 0, 0, K, K);
    if (rc) return rc;
    hipStream_t st = c->stream;
    if (const int rc_subs = upload_subsamples(c, spp, st)) return rc_subs;
    ptd::HitQueue hits{c->hit0.as<float4>(), c->hit1.as<uint32_t>()};
    std::vector<float4> h0, ro;
    std::vector<uint32_t> h1;
    for (uint64_t first = 0; first < total; first += K) {
        const uint32_t n_gen = static_cast<uint32_t>(std::min<uint64_t>(K, total - first));
        PT_HIP(c, hipMemsetAsync(c->counter.p, 0, 4, st));
        ptd::launch_generate(st, g.cam, c->subs.as<float2>(), sample, 1, 0, c->settings.seed, first, n_gen, 0,
                             ptd::TileSubset{0u, 1u, g.n_tiles}, c->pools[0].at(0), c->counter.as<uint32_t>());
        PT_HIP(c, hipMemcpyAsync(c->host_counter, c->counter.p, 4, hipMemcpyDeviceToHost, st));
        PT_SYNC(c, st);
        const uint32_t n = *c->host_counter;
        if (!n) continue;
        run_extend(c, st, c->pools[0].at(0), n, hits, nullptr);
        h0.resize(n);
        h1.resize(n);
        ro.resize(n);
        PT_HIP(c, hipMemcpyAsync(h0.data(), hits.h0, n * 16ull, hipMemcpyDeviceToHost, st));
        PT_HIP(c, hipMemcpyAsync(h1.data(), hits.h1, n * 4ull, hipMemcpyDeviceToHost, st));
        PT_HIP(c, hipMemcpyAsync(ro.data(), c->pools[0].o.p, n * 16ull, hipMemcpyDeviceToHost, st));
        PT_SYNC(c, st);
        std::vector<ptgpu_hit> dec(n);
        decode_hits(c, h0, h1, n, dec.data());
        for (uint32_t i = 0; i < n; ++i) {  // framebuffer slot (tile order) -> pixel
            uint32_t pid;
            std::memcpy(&pid, &ro[i].w, 4);
            const uint32_t tile = pid >> 6, lane = pid & 63u;
            const uint32_t x = (tile % g.cam.tiles_x) * ptd::kTile + (lane & 7u), y = (tile / g.cam.tiles_x) * ptd::kTile + (lane >> 3);
            out[static_cast<size_t>(y) * width + x] = dec[i];
        }
    }
    return PTGPU_OK;
}

int ptgpu_intersect_rays(ptgpu_ctx* c, uint64_t n, const float* origins, const float* directions, const uint8_t* any_hit,
                         ptgpu_hit* out) {
    if (!c || (n && (!origins || !directions || !out))) return PTGPU_E_INVALID;
    if (!c->scene_loaded) return c->fail(PTGPU_E_STATE, "intersect_rays: no scene loaded");
    PT_HIP(c, hipSetDevice(c->device));
    // rays per pass: 4 Mi unless ptgpu_options.intersect_chunk says otherwise (experiments with launch-sized passes)
    const uint64_t cap = c->options.intersect_chunk ? std::max<uint64_t>(c->options.intersect_chunk, 64) : (1u << 22);
    const uint64_t K = std::max<uint64_t>(std::min<uint64_t>(std::min<uint64_t>(c->chunk, cap), n), 64);
    int rc = ensure_work_memory(c, 0, 0, K, K);
    if (rc) return rc;
    hipStream_t st = c->stream;
    ptd::HitQueue hits{c->hit0.as<float4>(), c->hit1.as<uint32_t>()};
    DevBuf d_o, d_d, d_a;
    PT_HIP(c, d_o.ensure(K * 12));
    PT_HIP(c, d_d.ensure(K * 12));
    PT_HIP(c, d_a.ensure(K));
    std::vector<float4> h0;
    std::vector<uint32_t> h1;
    for (uint64_t first = 0; first < n; first += K) {
        const uint32_t m = static_cast<uint32_t>(std::min<uint64_t>(K, n - first));
        PT_HIP(c, hipMemcpyAsync(d_o.p, origins + 3 * first, m * 12ull, hipMemcpyHostToDevice, st));
        PT_HIP(c, hipMemcpyAsync(d_d.p, directions + 3 * first, m * 12ull, hipMemcpyHostToDevice, st));
        if (any_hit) PT_HIP(c, hipMemcpyAsync(d_a.p, any_hit + first, m, hipMemcpyHostToDevice, st));
        ptd::launch_pack_rays(st, d_o.as<float>(), d_d.as<float>(), any_hit ? d_a.as<uint8_t>() : nullptr, m, c->pools[0].at(0));
        run_extend(c, st, c->pools[0].at(0), m, hits, nullptr);
        h0.resize(m);
        h1.resize(m);
        PT_HIP(c, hipMemcpyAsync(h0.data(), hits.h0, m * 16ull, hipMemcpyDeviceToHost, st));
        PT_HIP(c, hipMemcpyAsync(h1.data(), hits.h1, m * 4ull, hipMemcpyDeviceToHost, st));
        PT_SYNC(c, st);
        decode_hits(c, h0, h1, m, out + first);
    }
    return PTGPU_OK;
}

int ptgpu_intersect_rays_static(ptgpu_ctx* c, uint32_t n, const float* d_origins, const float* d_directions,
                                const uint8_t* d_any_hit, ptgpu_hit* d_out, void* stream, int wait) {
    if (!c || (n && (!d_origins || !d_directions || !d_out))) return PTGPU_E_INVALID;
    if (!c->scene_loaded) return c->fail(PTGPU_E_STATE, "intersect_rays_static: no scene loaded");
    if (!c->static_scratch) return c->fail(PTGPU_E_STATE, "intersect_rays_static: device %d has no built-in work array", c->device);
    if (n > kStaticMaxRays)
        return c->fail(PTGPU_E_INVALID, "intersect_rays_static: %u rays in one call, the built-in work array holds %u", n, kStaticMaxRays);
    const uint32_t spill_depth = spill_depth_of(c);
    if (spill_depth > kStaticSpillDepth)
        return c->fail(PTGPU_E_NOMEM, "intersect_rays_static: the scene's trees need %u spilled stack entries per thread, the built-in work array holds %u",
                       spill_depth, kStaticSpillDepth);
    if (!n) return PTGPU_OK;
    PT_HIP(c, hipSetDevice(c->device));
    (void)ptd::take_launch_error(nullptr);
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : c->stream;
    unsigned char* base = c->static_scratch;  // the array's address on this device (never the symbol's name)
    const ptd::RayQueue rq{reinterpret_cast<float4*>(base + StaticLayout::ray_o), reinterpret_cast<float4*>(base + StaticLayout::ray_d),
                           reinterpret_cast<float4*>(base + StaticLayout::ray_p)};
    const ptd::HitQueue hits{reinterpret_cast<float4*>(base + StaticLayout::hit0), reinterpret_cast<uint32_t*>(base + StaticLayout::hit1)};
    const ptd::CandQueue cand{reinterpret_cast<uint2*>(base + StaticLayout::cand), kStaticMaxRays};
    ptd::launch_pack_rays(st, d_origins, d_directions, d_any_hit, n, rq);
    // the cursors and the overflow count are cleared by the launcher (hipMemsetAsync on these addresses); every ray's hit
    // record and candidate list is rewritten before it is read, so nothing else of the last call's contents matters
    ptd::launch_extend_persistent(st, c->scene, rq, n, hits, cand, reinterpret_cast<uint32_t*>(base + StaticLayout::spill), spill_depth,
                                  kStaticBlocks, reinterpret_cast<uint32_t*>(base + StaticLayout::cursors),
                                  reinterpret_cast<uint32_t*>(base + StaticLayout::overflow), nullptr);
    ptd::launch_unpack_hits(st, hits, c->d_object_mesh.as<uint32_t>(), n, d_out);
    const char* kernel = "";
    const hipError_t le = ptd::take_launch_error(&kernel);
    if (le != hipSuccess) return c->fail(PTGPU_E_HIP, "launch of %s failed: %s", kernel, hipGetErrorString(le));
    if (wait) PT_HIP(c, hipStreamSynchronize(st));
    return PTGPU_OK;
}

int ptgpu_get_bvh_info(const ptgpu_ctx* c, uint32_t mesh_index, ptgpu_bvh_info* out) {
    if (!c || !out) return PTGPU_E_INVALID;
    if (!c->scene_loaded || mesh_index >= c->bvh_info.size()) return PTGPU_E_STATE;
    *out = c->bvh_info[mesh_index];
    return PTGPU_OK;
}

int ptgpu_subsample_positions(uint32_t spp, float* out_xy) {
    if (!out_xy || spp == 0) return PTGPU_E_INVALID;
    const std::vector<float> t = subsample_table(spp);
    std::memcpy(out_xy, t.data(), t.size() * 4);
    return PTGPU_OK;
}

}  // extern "C"
