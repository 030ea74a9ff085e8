// pt_kernels.h — host-callable launchers of the wavefront kernels (generate, extend, shade, resolve).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_device.h"

namespace ptd {

constexpr int kBlock = 256;          // threads per workgroup: 4 wavefronts of 64 lanes
#ifndef PT_STACK_LDS
#define PT_STACK_LDS 12
#endif
constexpr int kStackLds = PT_STACK_LDS;  // per-lane traversal stack entries kept in LDS; deeper entries spill to HBM
constexpr int kTopNodes = 192;       // BVH nodes every traversal workgroup keeps in LDS (12 KB)
constexpr uint32_t kXformLds = 16;   // instance transforms kept in LDS by the same kernel (2 KB)
constexpr int kTopNodes4 = 192;      // likewise for the four-wide form (also 64 B a node)
// the same three for the four-wide kernel that runs seven workgroups per CU (SceneDev::seven_waves)
#ifndef PT_XFORM_LDS7
#define PT_XFORM_LDS7 12
#endif
#ifndef PT_STACK_LDS7
#define PT_STACK_LDS7 10
#endif
#ifndef PT_TOP_NODES7
#define PT_TOP_NODES7 176
#endif
constexpr int kStackLds7 = PT_STACK_LDS7;
constexpr int kTopNodes7 = PT_TOP_NODES7;
constexpr uint32_t kXformLds7 = PT_XFORM_LDS7;
// the seven-wave kernel's form for scenes with up to twice that many instances: all their transforms in LDS, fewer cached nodes
constexpr int kTopNodes7Many = kTopNodes7 - 2 * static_cast<int>(kXformLds7);
inline int top_nodes7_for(uint32_t n_instances) { return (n_instances > kXformLds7 && n_instances <= 2u * kXformLds7) ? kTopNodes7Many : kTopNodes7; }
constexpr int kInstLds = 16;         // per-lane cached instance entry distances kept in LDS
constexpr int kTile = 8;             // framebuffer / camera-ray tiles are 8x8 pixels = one wavefront

// Structure-of-arrays ray records, 48 B per ray:
//   o = origin.xyz, framebuffer slot (uint)        d = direction.xyz, depth | any_hit << 8 (uint)
//   p = payload.rgb, sampler key (uint)
struct RayQueue {
    float4* o;
    float4* d;
    float4* p;
};
// Hit records, 20 B per ray: h0 = t, A, B, triangle index (uint); h1 = object | backface << 30 | hit << 31
struct HitQueue {
    float4* h0;
    uint32_t* h1;
};

// Candidate instances per ray in visiting order: e[k * stride + ray] = (entry distance bits, instance slot) for k >= 1.  The
// first candidate is one word, first()[ray] = its slot: its distance is never looked at (a ray that has no hit yet enters its
// first candidate whatever the distance), and the words live in the first half of what would be plane 0 of `e`.  The slot word
// of the last stored entry carries kCandLast unless the list is full (kCandMax entries: it may go on, a rescan continues it);
// a ray without candidates has first()[ray] = PT_NO_NODE.
#ifndef PT_CAND_MAX
#define PT_CAND_MAX 8
#endif
constexpr int kCandMax = PT_CAND_MAX;
constexpr uint32_t kCandLast = 0x40000000u;  // instance slots are below 2^30
#ifndef PT_TLAS_LINEAR_MAX
#define PT_TLAS_LINEAR_MAX 96
#endif
constexpr uint32_t kTlasLinearMax = PT_TLAS_LINEAR_MAX;  // up to this many instances every ray tests every bound; above, a tree over the bounds culls
constexpr uint32_t kTlasLds = 512;       // nodes of that tree kept in LDS by the instance pass (16 KB)
struct CandQueue {
    uint2* e;
    uint32_t stride;
    uint32_t* h1 = nullptr;  // the launch's hit flag words (set by launch_extend_persistent): cleared by the instance pass
    __host__ __device__ uint32_t* first() const { return reinterpret_cast<uint32_t*>(e); }
};
static_assert(kCandMax >= 2, "the continuation pass reads the last entry with its distance");

struct CameraDev {
    float ndc2world[16];  // column-major
    float origin[3];
    float inv_w, inv_h;
    int32_t width, height;
    int32_t tiles_x, tiles_y;
};

// A pixel is kFrameBuckets accumulators per channel, one per range of magnitudes (six binades each, see add_radiance): a pixel
// of a 300-sample frame takes thousands of contributions over twelve orders of magnitude, and ONE binary32 accumulator drops
// every contribution below half a unit in the last place of the running sum — measured 3e-5 .. 7e-5 of the pixel's value at
// ~1000 rays per pixel, always downwards.  Contributions of like size add up among themselves without that loss; the
// buckets are added, smallest first, when the frame is resolved.
#ifndef PT_FRAME_BUCKETS
#define PT_FRAME_BUCKETS 8  // (1: one accumulator per channel, for comparisons)
#endif
constexpr uint32_t kFrameBuckets = PT_FRAME_BUCKETS;
struct FrameDev {
    float* sum;          // [bucket][R | G | B][stride] sums, indexed by framebuffer slot
    uint32_t stride;
    uint32_t* abandoned; // counts pass-through chains that were cut (see pass_hash in pt_kernels.hip); PBR only, else null
};

struct TraceCounters {   // filled by the counting variant of extend
    unsigned long long nodes, tris, inst_tested, inst_entered;
    unsigned long long wave_inner, wave_tri;  // wavefront-level loop iterations (SIMT utilisation = lane steps / 64 / these)
    // scheduler census of the persistent kernel: [phase 0 refill, 1 next, 2 inner, 3 leaf][0 executions, 1 steps,
    // 2..5 lanes idle / waiting for next / holding a node / holding a leaf, summed over the steps]
    unsigned long long census[4][6];
    unsigned long long leaf_records;          // 64-byte leaf records fetched (scenes that have them)
};

// slots_in_batch == 0: camera samples [first, first + n) of the linear index (sample_slot * n_tiles + tile) * 64 + lane.
// slots_in_batch == S: sample slots [first, first + S) of every tile in tile-major order, n = S * n_tiles * 64.
// The 8x8-pixel tiles a context renders: begin, begin + step, ... (count of them); {0, 1, all tiles} = the whole frame.
struct TileSubset {
    uint32_t begin, step, count;
};
// True when launch_generate writes ray i of a batch to slot i and the count it reports is the batch size (the frame is a
// whole number of tiles, so every index has a pixel).
bool generate_is_dense(const CameraDev& cam);
void launch_generate(hipStream_t st, const CameraDev& cam, const float2* subsamples, uint32_t sample_begin,
                     uint32_t sample_step, uint32_t max_bounces, uint32_t seed, uint64_t first, uint32_t n,
                     uint32_t slots_in_batch, TileSubset tiles, RayQueue out, uint32_t* out_count);
void launch_extend(hipStream_t st, const SceneDev& sc, RayQueue rays, uint32_t n, HitQueue hits, uint32_t* spill,
                   uint32_t spill_depth, uint32_t spill_stride, TraceCounters* counters);
// Workgroups that are resident at once for the persistent extend kernel (occupancy x compute units).
uint32_t extend_persistent_grid(int device, bool seven_waves);
// Persistent form: a fixed grid of n_blocks workgroups pulls rays from 8 cursors, one per XCD segment of the queue,
// 128 bytes apart at next_ray (1 KiB, zeroed here).
// `overflow` (1 + n words) lists the rays handed to the continuation pass (more than kCandMax candidate instances).
// `between`, when given, is recorded after the instance pass and before the traversal kernel.
void launch_extend_persistent(hipStream_t st, const SceneDev& sc, RayQueue rays, uint32_t n, HitQueue hits, CandQueue cand,
                              uint32_t* spill, uint32_t spill_depth, uint32_t n_blocks, uint32_t* next_ray, uint32_t* overflow,
                              TraceCounters* counters, hipEvent_t between = nullptr);
// program: 0 raycaster, 1 ambient occlusion, 2 PBR.  Appends spawned rays to `out` at *out_count.
// single_path (PBR): at most one spawned ray per hit, chosen in proportion to its payload and reweighted.
// What the PBR kernel needs to reserve its output slots through several counters instead of one (see pt_kernels.hip):
// `cursors` = kShadeStreamWords words of device memory (the counters, the staged count, a failure flag; zeroed by the launcher),
// `staging` = a ray queue of `staging_capacity` records for the fix-up of the ragged end.  The output region must have
// room for the launch's rays + shade_stream_slack() slots.  After the launch *out_count holds the number of rays and the
// word at cursors[kShadeStreamWords - 1] is non-zero if the staging queue was too small: the rays are all there, *out_count
// is not, and close_ragged_end_piecewise finishes the job.
struct ShadeStreams {
    uint32_t* cursors = nullptr;
    RayQueue staging{};
    uint32_t staging_capacity = 0;
};
constexpr uint32_t kShadeStreamWords = 32 * 32 + 2;  // (room for up to 32 counters, 128 bytes apart, + staged count + failure flag)
uint32_t shade_stream_slack();
// Whether launch_shade reserves this launch's output through the streams (only then are the cursor words rewritten).
bool shade_reserves_through_streams(int program, uint32_t n, const ShadeStreams* streams);
// For a launch whose failure word came back set: closes the ragged end in pieces the staging queue can hold (synchronises
// the stream once per piece).  *total and *out_count = the launch's rays.
hipError_t close_ragged_end_piecewise(hipStream_t st, RayQueue out, uint32_t* out_count, const ShadeStreams& streams, uint32_t* total);
void launch_shade(hipStream_t st, int program, const SceneDev& sc, RayQueue rays, uint32_t n, HitQueue hits, RayQueue out,
                  uint32_t* out_count, FrameDev fb, uint32_t ao_samples, bool single_path = false,
                  const ShadeStreams* streams = nullptr);
void launch_clear(hipStream_t st, float* p, uint64_t n);
// planar tiled sums -> row-major RGBA: rgb = sum / divisor (samples for the mean, 1 for raw sums), a = alpha
void launch_resolve(hipStream_t st, FrameDev fb, const CameraDev& cam, float* out_rgba, float divisor, float alpha);
// row-major RGBA sums -> means in place: rgb /= divisor, a = 1
void launch_divide_rgba(hipStream_t st, float* rgba, uint64_t n_pixels, float divisor);
// Hit queue records of rays [0, n) as the ABI's 32-byte ptgpu_hit records, written to device memory.
// object_mesh[object] = mesh index of an object.
void launch_unpack_hits(hipStream_t st, HitQueue hits, const uint32_t* object_mesh, uint32_t n, void* out_records);
// The first launch failure since the last call (hipSuccess if none) and the name of the kernel it happened to; clears it.
// Launchers return nothing, so callers ask here before they trust what a launch should have written.
hipError_t take_launch_error(const char** kernel);
void launch_pack_rays(hipStream_t st, const float* origins, const float* directions, const uint8_t* any_hit, uint32_t n,
                      RayQueue out);

}  // namespace ptd
