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

    // scene
    DevBuf nodes, nodes4, tlas, tris, instances, xforms, objects, corners, texels, env, lut;
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
    bool persistent = true;        // extend kernel form (PTGPU_EXTEND=simple selects one ray per lane)
    uint32_t persist_blocks = 0;   // resident workgroups of the persistent extend kernel
    uint32_t* host_counter = nullptr;  // pinned
    hipEvent_t ev[7] = {};

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

float srgb_to_linear_host(float ch) {  // cpu_framebuffer.h:21-27
    if (ch <= 0.04045f) return ch / 12.92f;
    return std::pow((ch + 0.055f) / 1.055f, 2.4f);
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

constexpr uint32_t kWideLeaf = 2;  // largest leaf of a tree built for four-wide nodes

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
    uint64_t cap = K + static_cast<uint64_t>(fanout) * K * gens;
    // at most 32 GiB per pool (PTGPU_POOL_GIB overrides); the scheduler splits segments beyond that
    static const uint64_t pool_gib = [] {
        const char* v = std::getenv("PTGPU_POOL_GIB");
        const long g = v ? std::atol(v) : 0;
        return static_cast<uint64_t>(g >= 1 ? g : 32);
    }();
    const uint64_t cap_limit = (pool_gib << 30) / 48;
    cap = std::min(cap, std::max<uint64_t>(cap_limit, K + static_cast<uint64_t>(fanout) * 64));
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
        if (const char* b = std::getenv("PTGPU_PERSIST_BLOCKS")) c->persist_blocks = static_cast<uint32_t>(std::max(1, std::atoi(b)));
        if (std::getenv("PTGPU_VERBOSE")) std::fprintf(stderr, "[ptgpu] persistent extend grid: %u workgroups\n", c->persist_blocks);
    }
    const uint64_t spill_columns = c->persistent ? static_cast<uint64_t>(c->persist_blocks) * ptd::kBlock : L;
    PT_HIP(c, c->spill.ensure(static_cast<size_t>(spill_depth) * spill_columns * 4 + 16));
    PT_HIP(c, c->ray_cursor.ensure(1024));
    if (c->persistent) PT_HIP(c, c->cand.ensure(static_cast<size_t>(ptd::kCandMax) * L * sizeof(uint2)));
    if (c->persistent) PT_HIP(c, c->overflow.ensure((L + 1) * sizeof(uint32_t)));
    PT_HIP(c, c->counter.ensure(16));
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
    if (per_slot <= c->chunk && n_slots) {
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
    PT_HIP(c, c->fb.ensure(static_cast<size_t>(g.fb_stride) * 3 * 4));
    ptd::FrameDev fb{c->fb.as<float>(), g.fb_stride};
    ptd::launch_clear(st, fb.sum, static_cast<uint64_t>(g.fb_stride) * 3);  // framebuffer.clear, renderer.cc:85
    const std::vector<float> sub = subsample_table(spp);
    PT_HIP(c, c->subs.ensure(sub.size() * 4));
    PT_HIP(c, hipMemcpyAsync(c->subs.p, sub.data(), sub.size() * 4, hipMemcpyHostToDevice, st));
    ptd::TraceCounters* d_tc = counted ? c->trace_counters.as<ptd::TraceCounters>() : nullptr;
    if (counted) PT_HIP(c, hipMemsetAsync(d_tc, 0, sizeof(ptd::TraceCounters), st));

    ptd::SceneDev sc = c->scene;
    ptd::HitQueue hits{c->hit0.as<float4>(), c->hit1.as<uint32_t>()};
    uint32_t* d_count = c->counter.as<uint32_t>();

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
    const char* dump_path = std::getenv("PTGPU_DUMP_RAYS");
    const uint64_t dump_launch = std::getenv("PTGPU_DUMP_LAUNCH") ? std::strtoull(std::getenv("PTGPU_DUMP_LAUNCH"), nullptr, 10) : 0;
    std::vector<Frame> frames;
    float ms = 0.0f;
    const bool verbose_launches = std::getenv("PTGPU_VERBOSE") && std::atoi(std::getenv("PTGPU_VERBOSE")) >= 2;
    // Rays carry their own depth, so generations may mix: when what a generation spawned still fits a launch together
    // with the next batch's camera rays, it is not launched on its own — the camera rays go on top of it in its pool.
    // Launches stay full that way (a generation shrinks from bounce to bounce; the alpha pass-through chains at the
    // end of a batch are slivers of a few thousand rays), and every launch saved is a tail saved.
    const uint64_t carry_below = (L > K) ? L - K + 1 : K / 32;
    int gen_pool = 0;  // pool that receives the next batch's camera rays (the one holding carried rays, if any)
    c->pools[0].top = c->pools[1].top = 0;
    for (uint64_t first = 0; first < total; first += K) {
        const uint32_t n_gen = static_cast<uint32_t>(std::min<uint64_t>(K, total - first));
        const bool last_batch = first + K >= total;
        Pool& gp = c->pools[gen_pool];
        if (gp.top + n_gen > gp.cap) return c->fail(PTGPU_E_NOMEM, "ray pool too small for a batch of camera rays");
        PT_HIP(c, hipMemsetAsync(d_count, 0, 4, st));
        PT_HIP(c, hipEventRecord(c->ev[0], st));
        if (slots_per_batch) {
            const uint32_t slot0 = static_cast<uint32_t>(first / per_slot);
            ptd::launch_generate(st, g.cam, c->subs.as<float2>(), begin, step, bounces, s.seed, slot0, n_gen,
                                 static_cast<uint32_t>(n_gen / per_slot), tiles, gp.at(gp.top), d_count);
        } else {
            ptd::launch_generate(st, g.cam, c->subs.as<float2>(), begin, step, bounces, s.seed, first, n_gen, 0, tiles, gp.at(gp.top),
                                 d_count);
        }
        PT_HIP(c, hipEventRecord(c->ev[1], st));
        PT_HIP(c, hipMemcpyAsync(c->host_counter, d_count, 4, hipMemcpyDeviceToHost, st));
        PT_SYNC(c, st);
        PT_HIP(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
        out.generate_ms += ms;
        const uint32_t n_cam = *c->host_counter;
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
                const uint64_t room = (op.cap - op.top) / fanout;
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
            if (dump_path && out.extend_launches == dump_launch) {  // developer aid: one launch's rays as raw float4 o[n], d[n]
                std::vector<float4> buf(2ull * n);
                PT_HIP(c, hipMemcpy(buf.data(), rq.o, n * 16ull, hipMemcpyDeviceToHost));
                PT_HIP(c, hipMemcpy(buf.data() + n, rq.d, n * 16ull, hipMemcpyDeviceToHost));
                if (FILE* f = std::fopen(dump_path, "wb")) {
                    std::fwrite(buf.data(), 16, buf.size(), f);
                    std::fclose(f);
                }
            }
            PT_HIP(c, hipEventRecord(c->ev[0], st));
            run_extend(c, st, rq, n, hits, d_tc, c->ev[6]);
            PT_HIP(c, hipEventRecord(c->ev[1], st));
            if (fanout) PT_HIP(c, hipMemsetAsync(d_count, 0, 4, st));
            PT_HIP(c, hipEventRecord(c->ev[2], st));
            ptd::launch_shade(st, program, sc, rq, n, hits, op.at(op.top), d_count, fb, ao_samples,
                              (s.flags & PTGPU_FLAG_SINGLE_PATH) != 0);
            PT_HIP(c, hipEventRecord(c->ev[3], st));
            if (fanout) PT_HIP(c, hipMemcpyAsync(c->host_counter, d_count, 4, hipMemcpyDeviceToHost, st));
            PT_SYNC(c, st);
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
    PT_SYNC(c, st);
    PT_HIP(c, hipEventElapsedTime(&ms, c->ev[4], c->ev[5]));
    out.total_ms = ms;
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
        if (std::getenv("PTGPU_VERBOSE")) {
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
    if (const char* m = std::getenv("PTGPU_EXTEND")) ctx->persistent = std::strcmp(m, "simple") != 0;
    if (const char* k = std::getenv("PTGPU_CHUNK")) {
        long v = std::atol(k);
        if (v >= 1024) ctx->chunk = static_cast<uint32_t>(std::min<long>(v, 1l << 28)) & ~63u;
    }
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
int ptgpu_get_render_settings(const ptgpu_ctx* c, ptgpu_settings* out) {
    if (!c || !out) return PTGPU_E_INVALID;
    *out = c->settings;
    return PTGPU_OK;
}

int ptgpu_load_scene(ptgpu_ctx* c, const ptgpu_model* model, const ptgpu_envmap* env) {
    if (!c || !model) return PTGPU_E_INVALID;
    PT_HIP(c, hipSetDevice(c->device));
    c->scene_loaded = false;
    // ---- validate what the kernels index with ----
    for (uint32_t i = 0; i < model->n_meshes; ++i) {
        const ptgpu_mesh& m = model->meshes[i];
        if (m.n_indices % 3 != 0) return c->fail(PTGPU_E_INVALID, "mesh %u: index count %u is not a multiple of 3", i, m.n_indices);
        if (static_cast<uint64_t>(m.first_vertex) + m.n_vertices > model->n_vertices ||
            static_cast<uint64_t>(m.first_index) + m.n_indices > model->n_indices)
            return c->fail(PTGPU_E_INVALID, "mesh %u: vertex/index range outside the model arrays", i);
        if (m.material_index >= model->n_materials) return c->fail(PTGPU_E_INVALID, "mesh %u: material index out of range", i);
        for (uint32_t k = 0; k < m.n_indices; ++k)
            if (model->indices[m.first_index + k] >= m.n_vertices)
                return c->fail(PTGPU_E_INVALID, "mesh %u: index %u out of range", i, k);
    }
    for (uint32_t i = 0; i < model->n_objects; ++i)
        if (model->objects[i].mesh_index >= model->n_meshes) return c->fail(PTGPU_E_INVALID, "object %u: mesh index out of range", i);
    if (model->n_objects >= (1u << 30)) return c->fail(PTGPU_E_INVALID, "too many objects");
    for (uint32_t i = 0; i < model->n_materials; ++i) {
        const ptgpu_material& m = model->materials[i];
        const int32_t tex[5] = {m.base_color_texture, m.metallic_roughness_texture, m.normal_texture, m.transmission_texture,
                                m.emissive_texture};
        for (int32_t t : tex)
            if (t >= static_cast<int32_t>(model->n_images)) return c->fail(PTGPU_E_INVALID, "material %u: texture index out of range", i);
    }
    for (uint32_t i = 0; i < model->n_images; ++i) {
        const ptgpu_image& im = model->images[i];
        if (im.width <= 0 || im.height <= 0 || im.texel_offset + static_cast<uint64_t>(im.width) * im.height > model->n_texels)
            return c->fail(PTGPU_E_INVALID, "image %u: bad size or texel range", i);
    }
    if (model->n_texels >= (1ull << 32)) return c->fail(PTGPU_E_INVALID, "texel pool too large");

    const ptgpu_settings s = c->settings;
    const bool naive = s.accel_struct_type == PTGPU_AS_NAIVE;
    // Four-wide nodes are for throughput, not for the reference's visiting order: their trees get small leaves (fewer
    // triangle tests per ray, a few more node visits; measured best at 2), whatever maxTrianglesPerBVHLeaf says.
    const bool want_wide = (s.flags & PTGPU_FLAG_WIDE_NODES) && c->persistent && !naive;
    const uint32_t leaf_setting = s.max_triangles_per_bvh_leaf ? s.max_triangles_per_bvh_leaf : 1;
    const uint32_t max_leaf = want_wide ? std::min(leaf_setting, kWideLeaf) : leaf_setting;
    // ... and are collapsed from a tree of lower expected cost than the reference's (2 % fewer node visits per ray);
    // PTGPU_WIDE_SPLITS=reference keeps the reference's splits under the wide nodes.
    const char* splits_env = std::getenv("PTGPU_WIDE_SPLITS");
    const bool tuned_splits = want_wide && !(splits_env && std::strcmp(splits_env, "reference") == 0);

    // ---- acceleration structure (BVH_AS::BVH_AS, acceleration_structure.cc:352-390) ----
    pth::InstanceTable table;
    pth::build_instances(*model, table);
    std::vector<ptd::DevNode> all_nodes;
    std::vector<ptd::DevTri> all_tris;
    std::vector<uint32_t> mesh_root(model->n_meshes);
    c->bvh_info.assign(model->n_meshes, ptgpu_bvh_info{});
    c->max_depth = 0;
    for (uint32_t i = 0; i < model->n_meshes; ++i) {
        const ptgpu_mesh& m = model->meshes[i];
        pth::MeshTree tree;
        if (s.accel_struct_type == PTGPU_AS_LBVH) {
            std::string err;
            if (!pth::build_mesh_tree_lbvh(model->vertices + m.first_vertex, model->indices + m.first_index, m.n_indices, max_leaf,
                                           c->stream, tree, &err))
                return c->fail(PTGPU_E_HIP, "mesh %u: %s", i, err.c_str());
        } else {
            pth::build_mesh_tree(model->vertices + m.first_vertex, model->indices + m.first_index, m.n_indices, max_leaf, naive,
                                 tuned_splits, tree);
        }
        if (all_tris.size() + tree.tris.size() >= (1ull << 31) - 2 || all_nodes.size() + tree.nodes.size() >= (1ull << 31))
            return c->fail(PTGPU_E_INVALID, "scene exceeds 2^31 triangles or nodes");
        pth::relocate(tree, static_cast<uint32_t>(all_nodes.size()), static_cast<uint32_t>(all_tris.size()));
        mesh_root[i] = tree.root;
        c->bvh_info[i] = tree.info;
        c->max_depth = std::max(c->max_depth, tree.info.max_depth);
        all_nodes.insert(all_nodes.end(), tree.nodes.begin(), tree.nodes.end());
        all_tris.insert(all_tris.end(), tree.tris.begin(), tree.tris.end());
    }
    for (ptd::DevXform& xf : table.xforms) {
        xf.root = mesh_root[xf.mesh_index];
        const ptgpu_material& mat = model->materials[model->meshes[xf.mesh_index].material_index];
        const bool double_sided = mat.double_sided || mat.has_volume;  // :377-378
        xf.cull = double_sided ? 0u : 1u;
    }
    std::memcpy(c->scene_min, table.scene_min, 12);
    std::memcpy(c->scene_max, table.scene_max, 12);

    // The most visited nodes go first in the node array: the traversal kernel keeps nodes[0 .. n_top) in LDS, which
    // takes their fetches off the vector L1's address unit (the busiest unit of that kernel).  "Most visited" is
    // estimated the way the surface area heuristic does: a node is visited in proportion to the area of its box,
    // summed over the instances of its mesh (scaled by the instance's size).  Only the numbering changes.
    uint32_t n_top = 0;
    std::vector<float> mesh_weight(model->n_meshes, 0.0f);  // summed squared scale of a mesh's instances
    for (const ptd::DevXform& xf : table.xforms) {
        const float* m = xf.model;  // columns c0 c1 c2 (c3 = translation)
        const float det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[3] * (m[1] * m[8] - m[2] * m[7]) + m[6] * (m[1] * m[5] - m[2] * m[4]);
        const float scale = std::cbrt(std::fabs(det));
        mesh_weight[xf.mesh_index] += scale * scale;
    }
    if (!all_nodes.empty()) {
        auto half_area = [](const float* lo, const float* hi) {
            const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
            return dx * dy + dx * dz + dy * dz;
        };
        struct Entry {
            float priority;
            uint32_t node, mesh;
            bool operator<(const Entry& o) const { return priority < o.priority; }
        };
        std::priority_queue<Entry> queue;
        for (uint32_t i = 0; i < model->n_meshes; ++i)
            if (!(mesh_root[i] & PT_LEAF_BIT) && mesh_weight[i] > 0.0f) queue.push({std::numeric_limits<float>::max(), mesh_root[i], i});
        std::vector<uint32_t> order;  // old indices of the chosen nodes
        while (!queue.empty() && order.size() < static_cast<size_t>(ptd::kTopNodes)) {
            const Entry e = queue.top();
            queue.pop();
            order.push_back(e.node);
            const ptd::DevNode& nd = all_nodes[e.node];
            if (!(nd.left & PT_LEAF_BIT)) queue.push({half_area(nd.l_lo, nd.l_hi) * mesh_weight[e.mesh], nd.left, e.mesh});
            if (!(nd.right & PT_LEAF_BIT)) queue.push({half_area(nd.r_lo, nd.r_hi) * mesh_weight[e.mesh], nd.right, e.mesh});
        }
        n_top = static_cast<uint32_t>(order.size());
        std::vector<uint32_t> new_index(all_nodes.size(), PT_NO_NODE);
        for (uint32_t k = 0; k < n_top; ++k) new_index[order[k]] = k;
        uint32_t next = n_top;
        for (uint32_t i = 0; i < all_nodes.size(); ++i)
            if (new_index[i] == PT_NO_NODE) new_index[i] = next++;
        std::vector<ptd::DevNode> moved(all_nodes.size());
        for (uint32_t i = 0; i < all_nodes.size(); ++i) {
            ptd::DevNode n = all_nodes[i];
            if (!(n.left & PT_LEAF_BIT)) n.left = new_index[n.left];
            if (!(n.right & PT_LEAF_BIT)) n.right = new_index[n.right];
            moved[new_index[i]] = n;
        }
        all_nodes.swap(moved);
        for (ptd::DevXform& xf : table.xforms)
            if (!(xf.root & PT_LEAF_BIT)) xf.root = new_index[xf.root];
    }

    // ---- a tree over the instances' boxes when there are many of them (the instance pass then culls before it tests) ----
    // Balanced by count (median split along the widest axis of the box centres), so its depth is ceil(log2(n)).
    std::vector<ptd::TlasNode> tlas;
    const char* tlas_env = std::getenv("PTGPU_TLAS");  // "linear": every ray tests every bound whatever their number (cross-check)
    if (table.instances.size() > ptd::kTlasLinearMax && !(tlas_env && !std::strcmp(tlas_env, "linear"))) {
        std::vector<uint32_t> order(table.instances.size());
        for (uint32_t k = 0; k < order.size(); ++k) order[k] = k;
        // depth-first emission with an explicit work list: a range becomes a node, then its left half, then its right half;
        // `skip` = the index after the node's subtree = node + 2 * count - 1
        struct Range {
            uint32_t begin, end;
        };
        std::vector<Range> todo{{0u, static_cast<uint32_t>(order.size())}};
        while (!todo.empty()) {
            const Range r = todo.back();
            todo.pop_back();
            const uint32_t count = r.end - r.begin;
            ptd::TlasNode nd;
            float clo[3], chi[3];
            for (int a = 0; a < 3; ++a) {
                nd.lo[a] = clo[a] = std::numeric_limits<float>::max();
                nd.hi[a] = chi[a] = std::numeric_limits<float>::lowest();
            }
            for (uint32_t i = r.begin; i < r.end; ++i) {
                const float* d = table.instances[order[i]].dop;
                for (int a = 0; a < 3; ++a) {
                    nd.lo[a] = std::min(nd.lo[a], d[2 * a]);
                    nd.hi[a] = std::max(nd.hi[a], d[2 * a + 1]);
                    const float c = d[2 * a] + d[2 * a + 1];
                    clo[a] = std::min(clo[a], c);
                    chi[a] = std::max(chi[a], c);
                }
            }
            nd.skip = static_cast<uint32_t>(tlas.size()) + 2u * count - 1u;
            nd.inst = (count == 1) ? order[r.begin] : PT_NO_NODE;
            tlas.push_back(nd);
            if (count == 1) continue;
            int axis = 0;
            for (int a = 1; a < 3; ++a)
                if (chi[a] - clo[a] > chi[axis] - clo[axis]) axis = a;
            const uint32_t mid = r.begin + count / 2;
            std::nth_element(order.begin() + r.begin, order.begin() + mid, order.begin() + r.end, [&](uint32_t x, uint32_t y) {
                const float* dx = table.instances[x].dop;
                const float* dy = table.instances[y].dop;
                const float cx = dx[2 * axis] + dx[2 * axis + 1], cy = dy[2 * axis] + dy[2 * axis + 1];
                return cx < cy || (cx == cy && x < y);
            });
            todo.push_back({mid, r.end});    // popped after the whole left half
            todo.push_back({r.begin, mid});
        }
    }

    // ---- optional four-wide nodes (PTGPU_FLAG_WIDE_NODES): inner children pulled up into their parent, largest box first ----
    std::vector<ptd::Node4> wide;
    uint32_t n_top4 = 0;
    bool seven_waves = false;
    if (want_wide && !all_nodes.empty()) {
        struct Child {
            uint32_t ref;
            float lo[3], hi[3];
        };
        auto half_area = [](const Child& ch) {
            const float dx = ch.hi[0] - ch.lo[0], dy = ch.hi[1] - ch.lo[1], dz = ch.hi[2] - ch.lo[2];
            return dx * dy + dx * dz + dy * dz;
        };
        auto children_of = [&](uint32_t node, Child out[2]) {
            const ptd::DevNode& nd = all_nodes[node];
            out[0].ref = nd.left;
            out[1].ref = nd.right;
            std::memcpy(out[0].lo, nd.l_lo, 12); std::memcpy(out[0].hi, nd.l_hi, 12);
            std::memcpy(out[1].lo, nd.r_lo, 12); std::memcpy(out[1].hi, nd.r_hi, 12);
        };
        std::vector<uint32_t> depth_of;                               // per wide node
        uint32_t max_depth4 = 0;
        struct Todo {
            uint32_t binary, wide_index;
        };
        std::vector<Todo> todo;
        auto make = [&](uint32_t binary, uint32_t depth) -> uint32_t {
            const uint32_t w = static_cast<uint32_t>(wide.size());
            wide.emplace_back();
            depth_of.push_back(depth);
            max_depth4 = std::max(max_depth4, depth);
            todo.push_back({binary, w});
            return w;
        };
        std::vector<uint32_t> mesh_root4(model->n_meshes, PT_NO_NODE);
        std::vector<uint32_t> binary_root(model->n_meshes, PT_NO_NODE);
        for (const ptd::DevXform& xf : table.xforms) binary_root[xf.mesh_index] = xf.root;
        // Which binary nodes survive as four-wide nodes.  A ray pays one visit per surviving node whose box it enters, so
        // the expected number of visits is the summed surface area of the surviving nodes; that sum is minimised exactly by
        // dynamic programming over the binary tree (Ylitie, Karras, Laine: Efficient incoherent ray traversal on GPUs
        // through compressed wide BVHs, 2017, section 4.1): cost[n][k] = least area below n when n's subtree is
        // represented by at most k+1 roots (k = 0: n itself is a wide node), part[n][k] = how many of them its left child gets
        // (0 = the solution for one root fewer is as good).  PTGPU_WIDE_COLLAPSE=greedy selects the older rule instead
        // (always pull up the child with the largest box).
        const char* collapse_env = std::getenv("PTGPU_WIDE_COLLAPSE");
        const bool optimal_collapse = !(collapse_env && !std::strcmp(collapse_env, "greedy"));
        std::vector<float> cost;
        std::vector<uint8_t> part;
        auto cost_of = [&](uint32_t ref, int roots) { return (ref & PT_LEAF_BIT) ? 0.0f : cost[static_cast<size_t>(ref) * 4 + (roots - 1)]; };
        if (optimal_collapse) {
            cost.assign(all_nodes.size() * 4, 0.0f);
            part.assign(all_nodes.size() * 4, 0);
            std::vector<uint32_t> order;  // every node after its children: reverse of a pre-order walk
            for (uint32_t m = 0; m < model->n_meshes; ++m) {
                if (binary_root[m] == PT_NO_NODE || (binary_root[m] & PT_LEAF_BIT)) continue;
                const size_t first = order.size();
                order.push_back(binary_root[m]);
                for (size_t i = first; i < order.size(); ++i) {
                    const ptd::DevNode& nd = all_nodes[order[i]];
                    if (!(nd.left & PT_LEAF_BIT)) order.push_back(nd.left);
                    if (!(nd.right & PT_LEAF_BIT)) order.push_back(nd.right);
                }
            }
            for (size_t i = order.size(); i-- > 0;) {
                const uint32_t n = order[i];
                const ptd::DevNode& nd = all_nodes[n];
                float lo[3], hi[3];
                for (int a = 0; a < 3; ++a) {
                    lo[a] = std::min(nd.l_lo[a], nd.r_lo[a]);
                    hi[a] = std::max(nd.l_hi[a], nd.r_hi[a]);
                }
                const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
                const float own = dx * dy + dx * dz + dy * dz;
                float below[5] = {0, 0, 0, 0, 0};  // below[k]: children represented by at most k roots, k = 2..4
                uint8_t left_share[5] = {0, 0, 0, 0, 0};
                for (int k = 2; k <= 4; ++k) {
                    below[k] = std::numeric_limits<float>::infinity();
                    for (int l = 1; l < k; ++l) {
                        const float c = cost_of(nd.left, l) + cost_of(nd.right, k - l);
                        if (c < below[k]) {
                            below[k] = c;
                            left_share[k] = static_cast<uint8_t>(l);
                        }
                    }
                }
                float* c = &cost[static_cast<size_t>(n) * 4];
                uint8_t* p4 = &part[static_cast<size_t>(n) * 4];
                c[0] = own + below[4];       // one root: n is a wide node, its (at most four) children come from below[4]
                p4[0] = left_share[4];
                for (int k = 2; k <= 4; ++k) {  // k roots: n dissolves into its children's roots, unless k - 1 roots do as well
                    if (c[k - 2] <= below[k]) {
                        c[k - 1] = c[k - 2];
                        p4[k - 1] = 0;
                    } else {
                        c[k - 1] = below[k];
                        p4[k - 1] = left_share[k];
                    }
                }
            }
        }
        for (uint32_t m = 0; m < model->n_meshes; ++m) {
            const uint32_t root = binary_root[m];
            if (root == PT_NO_NODE) continue;  // mesh without an instance
            if (root & PT_LEAF_BIT) {
                mesh_root4[m] = root;
                continue;
            }
            mesh_root4[m] = make(root, 1);
            while (!todo.empty()) {
                const Todo t = todo.back();
                todo.pop_back();
                Child ch[4];
                int n_ch = 2;
                children_of(t.binary, ch);
                if (optimal_collapse) {
                    // follow the recorded decisions: `roots` of the subtree below a child become children of this wide node
                    n_ch = 0;
                    struct Open {
                        Child c;
                        int roots;
                    };
                    Open open[8];
                    int n_open = 0;
                    Child two[2];
                    children_of(t.binary, two);
                    const int l = part[static_cast<size_t>(t.binary) * 4];
                    open[n_open++] = {two[1], 4 - l};
                    open[n_open++] = {two[0], l};
                    while (n_open) {
                        const Open o = open[--n_open];
                        if ((o.c.ref & PT_LEAF_BIT) || o.roots == 1) {
                            ch[n_ch++] = o.c;
                            continue;
                        }
                        const int share = part[static_cast<size_t>(o.c.ref) * 4 + (o.roots - 1)];
                        if (share == 0) {
                            open[n_open++] = {o.c, o.roots - 1};
                            continue;
                        }
                        children_of(o.c.ref, two);
                        open[n_open++] = {two[1], o.roots - share};
                        open[n_open++] = {two[0], share};
                    }
                }
                while (!optimal_collapse && n_ch < 4) {  // pull up the inner child with the largest box
                    int best = -1;
                    float best_area = -1.0f;
                    for (int k = 0; k < n_ch; ++k)
                        if (!(ch[k].ref & PT_LEAF_BIT) && half_area(ch[k]) > best_area) {
                            best_area = half_area(ch[k]);
                            best = k;
                        }
                    if (best < 0) break;
                    Child two[2];
                    children_of(ch[best].ref, two);
                    ch[best] = two[0];
                    ch[n_ch++] = two[1];
                }
                ptd::Node4 nd;
                std::memset(&nd, 0, sizeof(nd));
                // the node's own box, and per axis the step that reaches its far side in 255 steps (plane = origin + step * q,
                // evaluated in binary32 as the kernel's slow path does; low planes round down, high planes up)
                for (int a = 0; a < 3; ++a) {
                    float lo = ch[0].lo[a], hi = ch[0].hi[a];
                    for (int k = 1; k < n_ch; ++k) {
                        lo = std::min(lo, ch[k].lo[a]);
                        hi = std::max(hi, ch[k].hi[a]);
                    }
                    float step = (hi - lo) / 255.0f;
                    if (!(step > 0.0f) || !(step < std::numeric_limits<float>::max())) step = 0.0f;
                    // (far from the origin one unit in the last place of `hi` is many of `step`: grow geometrically)
                    for (float grow = 1.0f / 1024.0f; step > 0.0f && lo + step * 255.0f < hi; grow *= 2.0f)
                        step = std::max(std::nextafter(step, std::numeric_limits<float>::max()), step * (1.0f + grow));
                    nd.origin[a] = lo;
                    nd.scale[a] = step;
                    for (int k = 0; k < n_ch; ++k) {
                        int ql = 0, qh = 0;
                        if (step > 0.0f) {
                            ql = std::clamp(static_cast<int>(std::floor((ch[k].lo[a] - lo) / step)), 0, 255);
                            while (ql > 0 && lo + step * static_cast<float>(ql) > ch[k].lo[a]) --ql;
                            qh = std::clamp(static_cast<int>(std::ceil((ch[k].hi[a] - lo) / step)), 0, 255);
                            while (qh < 255 && lo + step * static_cast<float>(qh) < ch[k].hi[a]) ++qh;
                        }
                        nd.qlo[a] |= static_cast<uint32_t>(ql) << (8 * k);
                        nd.qhi[a] |= static_cast<uint32_t>(qh) << (8 * k);
                    }
                }
                // An unused slot is not marked "absent": it names the padding record after the last triangle (one degenerate
                // triangle, never hit) as a leaf, inside an inverted box (low planes 255, high planes 0) that no ray enters —
                // except in a node of zero extent, where the degenerate triangle is tested and missed.  The kernel so needs no
                // test for a slot's existence.
                const uint32_t dummy_leaf = PT_LEAF_BIT | static_cast<uint32_t>(all_tris.size());
                for (int k = n_ch; k < 4; ++k)
                    for (int a = 0; a < 3; ++a) {
                        nd.qlo[a] |= 255u << (8 * k);
                        nd.qhi[a] &= ~(255u << (8 * k));
                    }
                for (int k = 0; k < 4; ++k)
                    nd.ref[k] = (k >= n_ch) ? dummy_leaf
                                            : ((ch[k].ref & PT_LEAF_BIT) ? ch[k].ref : make(ch[k].ref, depth_of[t.wide_index] + 1));
                wide[t.wide_index] = nd;
            }
        }
        for (ptd::DevXform& xf : table.xforms) xf.root4 = mesh_root4[xf.mesh_index];
        // Which form of the four-wide kernel: seven workgroups per CU when nodes and triangles fit the Infinity Cache with room
        // to spare (the kernel is then limited by the latency of a step), six otherwise (limited by the gather path: more
        // wavefronts only add to it).  PTGPU_TRAVERSE_WAVES=6 / 7 overrides.
        {
            const uint64_t geometry_bytes = static_cast<uint64_t>(wide.size()) * sizeof(ptd::Node4) + (all_tris.size() + 2) * sizeof(ptd::TriQ);
            seven_waves = geometry_bytes <= (128ull << 20);
            if (const char* e = std::getenv("PTGPU_TRAVERSE_WAVES")) seven_waves = std::atoi(e) == 7;
        }
        const size_t top_capacity4 = seven_waves ? ptd::top_nodes7_for(static_cast<uint32_t>(table.instances.size())) : ptd::kTopNodes4;
        // the most visited wide nodes first (same estimate as for the binary nodes), kept in LDS by the kernel
        {
            struct Entry {
                float priority;
                uint32_t node, mesh;
                bool operator<(const Entry& o) const { return priority < o.priority; }
            };
            std::priority_queue<Entry> queue;
            for (uint32_t m = 0; m < model->n_meshes; ++m)
                if (mesh_root4[m] != PT_NO_NODE && !(mesh_root4[m] & PT_LEAF_BIT) && mesh_weight[m] > 0.0f)
                    queue.push({std::numeric_limits<float>::max(), mesh_root4[m], m});
            std::vector<uint32_t> order;
            while (!queue.empty() && order.size() < top_capacity4) {
                const Entry e = queue.top();
                queue.pop();
                order.push_back(e.node);
                const ptd::Node4& nd = wide[e.node];
                for (int k = 0; k < 4; ++k)
                    if (nd.ref[k] != PT_NO_NODE && !(nd.ref[k] & PT_LEAF_BIT)) {
                        float ext[3];
                        for (int a = 0; a < 3; ++a)
                            ext[a] = nd.scale[a] * static_cast<float>(static_cast<int>((nd.qhi[a] >> (8 * k)) & 0xffu) -
                                                                      static_cast<int>((nd.qlo[a] >> (8 * k)) & 0xffu));
                        queue.push({(ext[0] * ext[1] + ext[0] * ext[2] + ext[1] * ext[2]) * mesh_weight[e.mesh], nd.ref[k], e.mesh});
                    }
            }
            n_top4 = static_cast<uint32_t>(order.size());
            std::vector<uint32_t> new_index(wide.size(), PT_NO_NODE);
            for (uint32_t k = 0; k < n_top4; ++k) new_index[order[k]] = k;
            uint32_t next = n_top4;
            for (uint32_t i = 0; i < wide.size(); ++i)
                if (new_index[i] == PT_NO_NODE) new_index[i] = next++;
            std::vector<ptd::Node4> moved(wide.size());
            for (uint32_t i = 0; i < wide.size(); ++i) {
                ptd::Node4 n = wide[i];
                for (int k = 0; k < 4; ++k)
                    if (n.ref[k] != PT_NO_NODE && !(n.ref[k] & PT_LEAF_BIT)) n.ref[k] = new_index[n.ref[k]];
                moved[new_index[i]] = n;
            }
            wide.swap(moved);
            for (ptd::DevXform& xf : table.xforms)
                if (xf.root4 != PT_NO_NODE && !(xf.root4 & PT_LEAF_BIT)) xf.root4 = new_index[xf.root4];
        }
        // a visit pushes up to three children: the stack can hold 3 entries per level
        c->max_depth = std::max(c->max_depth, 3u * max_depth4 + 1u);
        if (std::getenv("PTGPU_VERBOSE"))
            std::fprintf(stderr, "[ptgpu] four-wide nodes: %zu (from %zu binary), depth %u, kernel form: %d workgroups per CU\n", wide.size(),
                         all_nodes.size(), max_depth4, seven_waves ? 7 : 6);
    }

    // ---- shading data ----
    // corners: every triangle's three vertices de-indexed, in mesh order (the hit record's triangle index / 3)
    std::vector<uint64_t> mesh_tri_base(model->n_meshes);
    uint64_t n_shade_tris = 0;
    for (uint32_t i = 0; i < model->n_meshes; ++i) {
        mesh_tri_base[i] = n_shade_tris;
        n_shade_tris += model->meshes[i].n_indices / 3;
    }
    if (n_shade_tris >= (1ull << 32)) return c->fail(PTGPU_E_INVALID, "scene exceeds 2^32 triangles");
    std::vector<ptgpu_vertex> corners(static_cast<size_t>(n_shade_tris) * 3);
    for (uint32_t i = 0; i < model->n_meshes; ++i) {
        const ptgpu_mesh& m = model->meshes[i];
        ptgpu_vertex* dst = corners.data() + mesh_tri_base[i] * 3;
        const uint32_t n_corner = m.n_indices - m.n_indices % 3;
        for (uint32_t k = 0; k < n_corner; ++k) {
            const uint32_t v = model->indices[m.first_index + k];
            if (v >= m.n_vertices) return c->fail(PTGPU_E_INVALID, "mesh %u: index %u beyond its %u vertices", i, v, m.n_vertices);
            dst[k] = model->vertices[m.first_vertex + v];
        }
    }
    auto texture_of = [&](int32_t image, ptd::DevTexture& out) -> bool {
        out = ptd::DevTexture{0, 0, 0u};
        if (image < 0) return true;
        if (static_cast<uint32_t>(image) >= model->n_images) return false;
        const ptgpu_image& im = model->images[image];
        if (im.width <= 0 || im.height <= 0 || im.texel_offset + static_cast<uint64_t>(im.width) * im.height > model->n_texels ||
            im.texel_offset >= (1ull << 32))
            return false;
        out = ptd::DevTexture{im.width, im.height, static_cast<uint32_t>(im.texel_offset)};
        return true;
    };
    std::vector<ptd::DevObject> objects(model->n_objects);
    c->object_mesh.resize(model->n_objects);
    for (uint32_t i = 0; i < model->n_objects; ++i) {
        const ptgpu_object& o = model->objects[i];
        ptd::DevObject& d = objects[i];
        std::memset(&d, 0, sizeof(d));
        for (int col = 0; col < 4; ++col)
            for (int r = 0; r < 3; ++r) {
                d.model[col * 3 + r] = o.model_matrix[col * 4 + r];
                d.normal[col * 3 + r] = o.normal_matrix[col * 4 + r];
            }
        d.tri_base = static_cast<uint32_t>(mesh_tri_base[o.mesh_index]);
        c->object_mesh[i] = o.mesh_index;
        const ptgpu_material& m = model->materials[model->meshes[o.mesh_index].material_index];
        ptd::DevMaterial& dm = d.mat;
        std::memcpy(dm.base_color, m.base_color_factor, 16);
        std::memcpy(dm.emissive, m.emissive_factor, 12);
        dm.emissive_strength = m.emissive_strength;
        std::memcpy(dm.attenuation, m.attenuation_factor, 12);
        dm.metallic = m.metallic_factor;
        dm.roughness = m.roughness_factor;
        dm.ior = m.ior;
        dm.f0 = m.dielectric_f0;
        dm.transmission = m.transmission_factor;
        dm.alpha_cutoff = m.alpha_cutoff;
        dm.double_sided = m.double_sided != 0;
        dm.has_volume = m.has_volume != 0;
        dm.alpha_blending = m.alpha_blending != 0;
        if (!texture_of(m.base_color_texture, dm.base_color_tex) || !texture_of(m.metallic_roughness_texture, dm.mr_tex) ||
            !texture_of(m.normal_texture, dm.normal_tex) || !texture_of(m.transmission_texture, dm.transmission_tex) ||
            !texture_of(m.emissive_texture, dm.emissive_tex))
            return c->fail(PTGPU_E_INVALID, "object %u: its material names an image that does not exist or lies outside the texel array", i);
    }
    static_assert(sizeof(ptgpu_vertex) == 48, "vertex is three float4");
    static_assert(sizeof(ptd::DevObject) == 256, "object record is 256 bytes");
    float lut[256];
    for (int i = 0; i < 256; ++i) lut[i] = srgb_to_linear_host(static_cast<float>(i) / 255.0f);
    const float white[4] = {1.0f, 1.0f, 1.0f, 1.0f};  // CPUTexture<hdr_pixel>::create_white_texture, cpu_framebuffer.cc:104
    const float* env_px = white;
    int env_w = 1, env_h = 1;
    if (env && env->rgba) {
        if (env->width <= 0 || env->height <= 0) return c->fail(PTGPU_E_INVALID, "environment map has a bad size");
        env_px = env->rgba;
        env_w = env->width;
        env_h = env->height;
    }

    // traversal forms: planes grouped per axis, triangle edges precomputed (binary32 subtraction, as the test does)
    std::vector<ptd::NodeQ> q_nodes(all_nodes.size());
    for (size_t i = 0; i < all_nodes.size(); ++i) {
        const ptd::DevNode& n = all_nodes[i];
        ptd::NodeQ& q = q_nodes[i];
        float* axes[3] = {q.x, q.y, q.z};
        for (int a = 0; a < 3; ++a) {
            axes[a][0] = n.l_lo[a];
            axes[a][1] = n.r_lo[a];
            axes[a][2] = n.l_hi[a];
            axes[a][3] = n.r_hi[a];
        }
        q.left = n.left;
        q.right = n.right;
        q.pad0 = q.pad1 = 0;
    }
    // + two records of padding: the four-wide kernel reads triangles in pairs, and the first padding record is itself a leaf
    // (one degenerate triangle: the filler of unused child slots)
    std::vector<ptd::TriQ> q_tris(all_tris.size() + 2);
    for (size_t k = all_tris.size(); k < q_tris.size(); ++k) {
        std::memset(&q_tris[k], 0, sizeof(ptd::TriQ));
        q_tris[k].leaf_count = 1;
    }
    uint32_t in_leaf = 0;  // the builder marks a leaf's first triangle with the leaf size; every triangle gets the rest count
    for (size_t i = 0; i < all_tris.size(); ++i) {
        const ptd::DevTri& t = all_tris[i];
        ptd::TriQ& q = q_tris[i];
        for (int a = 0; a < 3; ++a) {
            q.p1[a] = t.p1[a];
            q.e1[a] = t.p2[a] - t.p1[a];
            q.e2[a] = t.p3[a] - t.p1[a];
        }
        q.index = t.index;
        if (t.leaf_count) in_leaf = t.leaf_count;
        q.leaf_count = in_leaf ? in_leaf-- : 1u;
        q.pad = 0;
    }
    int rc;
    if ((rc = upload(c, c->nodes, q_nodes.data(), q_nodes.size()))) return rc;
    if ((rc = upload(c, c->tris, q_tris.data(), q_tris.size()))) return rc;
    if (!wide.empty() && (rc = upload(c, c->nodes4, wide.data(), wide.size()))) return rc;
    if (!tlas.empty() && (rc = upload(c, c->tlas, tlas.data(), tlas.size()))) return rc;
    if ((rc = upload(c, c->instances, table.instances.data(), table.instances.size()))) return rc;
    if ((rc = upload(c, c->xforms, table.xforms.data(), table.xforms.size()))) return rc;
    if ((rc = upload(c, c->objects, objects.data(), objects.size()))) return rc;
    if ((rc = upload(c, c->corners, corners.data(), corners.size()))) return rc;
    if ((rc = upload(c, c->texels, reinterpret_cast<const uint32_t*>(model->texels), static_cast<size_t>(model->n_texels)))) return rc;
    if ((rc = upload(c, c->env, env_px, static_cast<size_t>(env_w) * env_h * 4))) return rc;
    if ((rc = upload(c, c->lut, lut, 256))) return rc;
    if ((rc = upload(c, c->d_object_mesh, c->object_mesh.data(), c->object_mesh.size()))) return rc;

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
    sc.n_instances = static_cast<uint32_t>(table.instances.size());
    sc.n_top_nodes = n_top;
    sc.nodes4 = wide.empty() ? nullptr : c->nodes4.as<ptd::Node4>();
    sc.tlas = tlas.empty() ? nullptr : c->tlas.as<ptd::TlasNode>();
    sc.n_tlas = static_cast<uint32_t>(tlas.size());
    sc.n_top_nodes4 = n_top4;
    sc.seven_waves = seven_waves ? 1u : 0u;
    {   // deep closest-hit traversals (large scenes, every program but ambient occlusion with its any-hit rays) give leaves the
        // heavier vote in the traversal kernel's choice of phase; PTGPU_LEAF_VOTE=<1..16> overrides
        uint64_t instanced = 0;
        for (uint32_t i = 0; i < model->n_objects; ++i) instanced += model->meshes[model->objects[i].mesh_index].n_indices / 3;
        sc.leaf_vote = (s.program_mode != PTGPU_PROGRAM_AO && instanced >= 500000) ? 4u : 1u;
        if (const char* v = std::getenv("PTGPU_LEAF_VOTE")) {
            const long k = std::atol(v);
            if (k >= 1 && k <= 16) sc.leaf_vote = static_cast<uint32_t>(k);
        }
    }
    c->persist_blocks = 0;  // the resident grid depends on the kernel form: sized again at the next render
    sc.env_w = env_w;
    sc.env_h = env_h;
    sc.env_rot = s.envmap_rotation;
    c->scene_settings = s;
    c->scene_loaded = true;
    return PTGPU_OK;
}

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
    const std::vector<float> sub = subsample_table(spp);
    PT_HIP(c, c->subs.ensure(sub.size() * 4));
    PT_HIP(c, hipMemcpyAsync(c->subs.p, sub.data(), sub.size() * 4, hipMemcpyHostToDevice, st));
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
    // rays per pass: 4 Mi unless PTGPU_INTERSECT_CHUNK says otherwise (experiments with launch-sized passes)
    const char* cap_env = std::getenv("PTGPU_INTERSECT_CHUNK");
    const uint64_t cap = cap_env ? std::max<uint64_t>(std::strtoull(cap_env, nullptr, 10), 64) : (1u << 22);
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
