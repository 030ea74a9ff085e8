// pt_kernels.hip — wavefront path-tracing kernels for gfx950 (MI355X).
//
//   generate  camera rays for a range of (sample, 8x8 tile, lane) indices          renderer.cc:17-21, 66-75
//   extend    two-level traversal: instance DOPs, per-mesh binary BVH, triangles   acceleration_structure.cc:587-708
//   shade     RayCaster / AmbientOcclusion / PBR programs; spawned rays are          ray_program.cc:150-391
//             compacted per wavefront with ballots and appended with one atomic
//   resolve   tiled planar sums -> row-major RGBA means                             renderer.cc:118-120
//
// (citations: files of the reference renderer whose arithmetic each kernel reproduces)
//
// One lane owns one ray.  Traversal stacks are per lane: the first kStackLds entries of every lane
// live in LDS as stack[entry][lane] (bank = lane % 32 for every entry, so pushes and pops never
// conflict whatever the lanes' stack heights); deeper entries go to a per-lane HBM spill area.
#include <type_traits>
#include "pt_kernels.h"

#include <algorithm>

namespace ptd {

namespace {

constexpr float kInfF = __builtin_inff();
constexpr float kPiF = 3.14159265358979323846264338327950288f;
constexpr float kInvSqrt3 = 0.577350269189625764509148780501957456f;

__device__ __forceinline__ f3 ld3(const float4& v) { return f3{v.x, v.y, v.z}; }

// ---------------------------------------------------------------------------------------------
// volumes
// ---------------------------------------------------------------------------------------------
// ray_volume_hit_info::forward_hit_distance (ray_program.h:44-47)
__device__ __forceinline__ float forward_distance(bool hit, float t0, float t1) {
    return (hit && (t1 > 0.0f)) ? (t0 > 0.0f ? t0 : 0.0f) : kInfF;
}
// BBox::ray_volume_intersection(ray, inv_dir) (acceleration_structure.cc:109-122)
__device__ __forceinline__ void box_test(float lox, float loy, float loz, float hix, float hiy, float hiz, f3 o, f3 inv,
                                         bool& hit, float& dist) {
    float ax = (lox - o.x) * inv.x, ay = (loy - o.y) * inv.y, az = (loz - o.z) * inv.z;
    float bx = (hix - o.x) * inv.x, by = (hiy - o.y) * inv.y, bz = (hiz - o.z) * inv.z;
    float t0x = sel_min(ax, bx), t0y = sel_min(ay, by), t0z = sel_min(az, bz);
    float t1x = sel_max(ax, bx), t1y = sel_max(ay, by), t1z = sel_max(az, bz);
    t0x = (t0x > t0y) ? t0x : t0y;
    t0x = (t0x > t0z) ? t0x : t0z;
    t1x = (t1x < t1y) ? t1x : t1y;
    t1x = (t1x < t1z) ? t1x : t1z;
    hit = (t0x <= t1x);
    dist = forward_distance(hit, t0x, t1x);
}
// Ray projections on the 7 DOP axes, cached per ray (acceleration_structure.cc:593-597; axes .h:50-58)
struct DopRay {
    float op[7], ip[7];
};
__device__ __forceinline__ DopRay dop_projections(f3 o, f3 d) {
    const f3 axes[7] = {f3{1.0f, 0.0f, 0.0f}, f3{0.0f, 1.0f, 0.0f}, f3{0.0f, 0.0f, 1.0f},
                        f3{kInvSqrt3, kInvSqrt3, kInvSqrt3}, f3{kInvSqrt3, kInvSqrt3, -kInvSqrt3},
                        f3{kInvSqrt3, -kInvSqrt3, kInvSqrt3}, f3{-kInvSqrt3, kInvSqrt3, kInvSqrt3}};
    DopRay r;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        r.op[i] = dot3(o, axes[i]);
        r.ip[i] = 1.0f / dot3(d, axes[i]);
    }
    return r;
}
// DOP::ray_volume_intersection(projections) + forward_hit_distance (acceleration_structure.cc:192-206)
__device__ __forceinline__ float dop_forward_distance(const float* __restrict__ dop, const DopRay& pr) {
    float tmin = -3.402823466e+38f, tmax = 3.402823466e+38f;
    bool hit = true;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        float t0 = (dop[2 * i] - pr.op[i]) * pr.ip[i];
        float t1 = (dop[2 * i + 1] - pr.op[i]) * pr.ip[i];
        if (t0 > t1) { float s = t0; t0 = t1; t1 = s; }
        tmin = (tmin < t0) ? t0 : tmin;  // std::max(t0, tmin)
        tmax = (t1 < tmax) ? t1 : tmax;  // std::min(t1, tmax)
        if (hit && (tmax < tmin)) hit = false;  // the reference returns here; later axes cannot un-miss
        // after the three box slabs most misses are decided: leave when no lane of the wavefront is still inside
        if (i == 2 && __ballot(hit) == 0ull) return kInfF;
    }
    return forward_distance(hit, tmin, tmax);
}

// The same test for rays whose seven inverse projections are all finite: no operand can be NaN then, so hardware
// min / max give what the compare-and-select forms above give, and "some slab interval is empty" can be decided once
// at the end (tmin only grows, tmax only shrinks).
__device__ __forceinline__ float dop_forward_distance_finite(const float* __restrict__ dop, const DopRay& pr) {
    float tmin = -3.402823466e+38f, tmax = 3.402823466e+38f;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const float t0 = (dop[2 * i] - pr.op[i]) * pr.ip[i];
        const float t1 = (dop[2 * i + 1] - pr.op[i]) * pr.ip[i];
        tmin = __builtin_fmaxf(tmin, __builtin_fminf(t0, t1));
        tmax = __builtin_fminf(tmax, __builtin_fmaxf(t0, t1));
        if (i == 2 && __ballot(tmin <= tmax) == 0ull) return kInfF;
    }
    return forward_distance(tmin <= tmax, tmin, tmax);
}

struct LaneHit {
    float t, A, B;
    uint32_t tri, object;
    bool hit, backface;
};

struct TraceCount {
    uint32_t nodes, tris, inst_tested, inst_entered;
    uint32_t wave_inner, wave_tri;  // loop iterations executed by the wavefront (counted by one lane per iteration)
    uint32_t census[4][6];
    uint32_t leaf_records;
};
__device__ __forceinline__ bool first_active_lane() {
    const unsigned long long m = __ballot(1);
    return (threadIdx.x & 63u) == static_cast<uint32_t>(__ffsll(static_cast<long long>(m)) - 1);
}

// Moller-Trumbore exactly as acceleration_structure.cc:16-49; returns true when the triangle is hit
// (e1 = p2 - p1 and e2 = p3 - p1 come precomputed with the triangle)
__device__ __forceinline__ bool ray_triangle(f3 o, f3 d, f3 p1, f3 e1, f3 e2, bool cull, float& A, float& B, float& t,
                                             bool& backface) {
    f3 pv = cross3(d, e2);
    float det = dot3(e1, pv);
    if (det == 0.0f) return false;
    if (cull && det < 0.0f) return false;
    float inv_det = 1.0f / det;
    f3 tv = o - p1;
    float u = dot3(tv, pv) * inv_det;
    if (u < 0.0f || u > 1.0f) return false;
    f3 qv = cross3(tv, e1);
    float v = dot3(d, qv) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return false;
    t = dot3(e2, qv) * inv_det;
    A = 1.0f - u - v;
    B = u;
    backface = (det < 0.0f) && !cull;
    return true;
}

// BVH_AS::mesh_ray_intersection (acceleration_structure.cc:635-708) for one instance.
//
// Every lane walks its own tree in the reference's order (nearer child first, farther child on the
// stack).  The loop is split "while-while": all lanes of the wavefront first descend through inner
// nodes until each of them holds a leaf (or is finished), then all leaf-holding lanes test their
// triangles together.  Each lane's own sequence of node visits and triangle tests is unchanged; only
// the interleaving across lanes differs, so the 8-triangle leaf loop runs once per round for the whole
// wave instead of once per step.
template <bool ANY, bool COUNT>
__device__ __forceinline__ void trace_instance(const SceneDev& sc, uint32_t slot, f3 o_ws, f3 d_ws, LaneHit& best,
                                               uint32_t (*stack)[kBlock], uint32_t* __restrict__ spill,
                                               uint32_t spill_depth, uint32_t spill_stride, uint32_t gid, TraceCount& cnt) {
    const DevXform& xf = sc.xforms[slot];
    const f3 o = xform(xf.inv, o_ws, 1.0f);
    const f3 d = normalize3(xform(xf.inv, d_ws, 0.0f));
    const f3 inv = f3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    const bool cull = xf.cull != 0;
    const uint32_t tid = threadIdx.x;

    float h_t = kInfF, h_A = 0.0f, h_B = 0.0f;
    uint32_t h_tri = 0;
    bool h_hit = false, h_back = false;

    int sp = 0;
    uint32_t cur = xf.root;  // PT_NO_NODE once this lane has nothing left to visit
    for (;;) {
        // ---- inner nodes: until this lane holds a leaf or is finished ----
        while (cur != PT_NO_NODE && !(cur & PT_LEAF_BIT)) {
            if (COUNT) { ++cnt.nodes; if (first_active_lane()) ++cnt.wave_inner; }
            const float4* np = reinterpret_cast<const float4*>(sc.nodes + cur);
            const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
            bool hl, hr;
            float dl, dr;
            box_test(n0.x, n1.x, n2.x, n0.z, n1.z, n2.z, o, inv, hl, dl);
            box_test(n0.y, n1.y, n2.y, n0.w, n1.w, n2.w, o, inv, hr, dr);
            const uint32_t left = __float_as_uint(n3.x), right = __float_as_uint(n3.y);
            const bool take_l = hl && (h_t > dl), take_r = hr && (h_t > dr);
            // the reference pushes the farther child first and pops the nearer one next; the nearer
            // child is kept in a register here instead of a push followed by a pop
            const bool l_first = dl < dr;
            const uint32_t near_ref = l_first ? left : right, far_ref = l_first ? right : left;
            const bool take_near = l_first ? take_l : take_r, take_far = l_first ? take_r : take_l;
            if (take_near) {
                if (take_far) {
                    if (sp < kStackLds) stack[sp][tid] = far_ref;
                    else if (static_cast<uint32_t>(sp - kStackLds) < spill_depth)
                        spill[static_cast<size_t>(sp - kStackLds) * spill_stride + gid] = far_ref;
                    ++sp;
                }
                cur = near_ref;
            } else if (take_far) {
                cur = far_ref;
            } else if (sp == 0) {
                cur = PT_NO_NODE;
            } else {
                --sp;
                cur = (sp < kStackLds) ? stack[sp][tid] : spill[static_cast<size_t>(sp - kStackLds) * spill_stride + gid];
            }
        }
        if (cur == PT_NO_NODE) break;
        // ---- leaf: test its triangles in order, the next triangle is fetched while one is tested ----
        {
            const float4* tp = reinterpret_cast<const float4*>(sc.tris + (cur & ~PT_LEAF_BIT));
            float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
            const uint32_t count = __float_as_uint(q2.z);
            for (uint32_t k = 0;;) {
                ++k;
                float4 r0 = q0, r1 = q1, r2 = q2;
                if (k < count) {
                    tp += 3;
                    r0 = tp[0]; r1 = tp[1]; r2 = tp[2];
                }
                if (COUNT) { ++cnt.tris; if (first_active_lane()) ++cnt.wave_tri; }
                float A, B, t;
                bool bf;
                const bool ok = ray_triangle(o, d, f3{q0.x, q0.y, q0.z}, f3{q0.w, q1.x, q1.y}, f3{q1.z, q1.w, q2.x}, cull,
                                             A, B, t, bf);
                if (ok && !(t < 0.0f) && !(t >= h_t)) {
                    h_A = A; h_B = B; h_t = t; h_hit = true; h_back = bf;
                    h_tri = __float_as_uint(q2.y);
                    if (ANY) break;
                }
                if (k >= count) break;
                q0 = r0; q1 = r1; q2 = r2;
            }
            if ((ANY && h_hit) || sp == 0) {
                cur = PT_NO_NODE;
            } else {
                --sp;
                cur = (sp < kStackLds) ? stack[sp][tid] : spill[static_cast<size_t>(sp - kStackLds) * spill_stride + gid];
            }
        }
    }
    if (!h_hit) return;
    // back to a world-space distance (acceleration_structure.cc:704-706)
    const f3 p = o + d * h_t;
    const f3 p_ws = xform(xf.model, p, 1.0f);
    const float t_ws = length3(p_ws - o_ws);
    // acceptance at the top level (acceleration_structure.cc:616-623)
    if ((t_ws > 0.0f) && (t_ws < best.t)) {
        best.t = t_ws; best.A = h_A; best.B = h_B; best.tri = h_tri; best.hit = true; best.backface = h_back;
        best.object = xf.object_index;
    }
}

// BVH_AS::intersect_ray (acceleration_structure.cc:587-633)
template <bool COUNT>
__device__ __forceinline__ LaneHit intersect_ray(const SceneDev& sc, f3 o, f3 d, bool any_hit, uint32_t (*stack)[kBlock],
                                                 float (*dist_cache)[kBlock], uint32_t* spill, uint32_t spill_depth,
                                                 uint32_t spill_stride, uint32_t gid, TraceCount& cnt) {
    LaneHit best;
    best.t = kInfF; best.A = 0.0f; best.B = 0.0f; best.tri = 0; best.object = 0; best.hit = false; best.backface = false;
    const uint32_t n_inst = sc.n_instances;
    const uint32_t tid = threadIdx.x;
    // entry distance of every instance's DOP, computed once; the first kInstLds are cached in LDS
    {
        const DopRay pr = dop_projections(o, d);
        for (uint32_t k = 0; k < n_inst && k < kInstLds; ++k) dist_cache[k][tid] = dop_forward_distance(sc.instances[k].dop, pr);
    }
    if (COUNT) cnt.inst_tested += n_inst;
    auto entry = [&](uint32_t k) -> float {
        if (k < kInstLds) return dist_cache[k][tid];
        return dop_forward_distance(sc.instances[k].dop, dop_projections(o, d));  // scenes with many objects: recomputed
    };
    if (any_hit) {
        // cheapest objects first: walk the complexity-sorted table from the back, stop at the first hit
        for (uint32_t k = n_inst; k-- > 0;) {
            if (entry(k) < kInfF) {
                if (COUNT) ++cnt.inst_entered;
                trace_instance<true, COUNT>(sc, k, o, d, best, stack, spill, spill_depth, spill_stride, gid, cnt);
                if (best.hit) break;
            }
        }
        return best;
    }
    // closest hit: visit instances in order of increasing DOP entry distance (ties: table order),
    // stop when the nearest unvisited entry is not in front of the current hit
    float last_d = -1.0f;
    uint32_t last_k = 0;
    for (;;) {
        float best_d = kInfF;
        uint32_t best_k = PT_NO_NODE;
        for (uint32_t k = 0; k < n_inst; ++k) {
            const float dist = entry(k);
            const bool after = (dist > last_d) || (dist == last_d && k > last_k);
            if (after && dist < best_d) { best_d = dist; best_k = k; }
        }
        if (best_k == PT_NO_NODE || !(best_d < best.t)) break;
        if (COUNT) ++cnt.inst_entered;
        trace_instance<false, COUNT>(sc, best_k, o, d, best, stack, spill, spill_depth, spill_stride, gid, cnt);
        last_d = best_d;
        last_k = best_k;
    }
    return best;
}

#ifdef PT_EXTEND_WAVES
#define PT_EXTEND_BOUNDS __launch_bounds__(kBlock, PT_EXTEND_WAVES)
#else
#define PT_EXTEND_BOUNDS __launch_bounds__(kBlock)
#endif
template <bool COUNT>
__global__ PT_EXTEND_BOUNDS void extend_kernel(SceneDev sc, const float4* __restrict__ ray_o,
                                                        const float4* __restrict__ ray_d, uint32_t n,
                                                        float4* __restrict__ hit0, uint32_t* __restrict__ hit1,
                                                        uint32_t* __restrict__ spill, uint32_t spill_depth,
                                                        uint32_t spill_stride, TraceCounters* counters) {
    __shared__ uint32_t s_stack[kStackLds][kBlock];
    __shared__ float s_dist[kInstLds][kBlock];
    const uint32_t gid = blockIdx.x * kBlock + threadIdx.x;
    TraceCount cnt{0, 0, 0, 0, 0, 0};
    if (gid < n) {
        const float4 ro = ray_o[gid], rd = ray_d[gid];
        const bool any_hit = ((__float_as_uint(rd.w) >> 8) & 1u) != 0;
        LaneHit h = intersect_ray<COUNT>(sc, ld3(ro), ld3(rd), any_hit, s_stack, s_dist, spill, spill_depth, spill_stride,
                                         gid, cnt);
        hit0[gid] = make_float4(h.t, h.A, h.B, __uint_as_float(h.tri));
        hit1[gid] = h.object | (h.backface ? 0x40000000u : 0u) | (h.hit ? 0x80000000u : 0u);
    }
    if (COUNT) {
        atomicAdd(&counters->nodes, static_cast<unsigned long long>(cnt.nodes));
        atomicAdd(&counters->tris, static_cast<unsigned long long>(cnt.tris));
        atomicAdd(&counters->inst_tested, static_cast<unsigned long long>(cnt.inst_tested));
        atomicAdd(&counters->inst_entered, static_cast<unsigned long long>(cnt.inst_entered));
        atomicAdd(&counters->wave_inner, static_cast<unsigned long long>(cnt.wave_inner));
        atomicAdd(&counters->wave_tri, static_cast<unsigned long long>(cnt.wave_tri));
    }
}

// ---------------------------------------------------------------------------------------------
// extend as two kernels: tlas (candidate instances per ray) + traverse (persistent BVH walk)
// ---------------------------------------------------------------------------------------------
// Rays differ by an order of magnitude in traversal length (a ray that misses every DOP costs nothing,
// a ray grazing a million-triangle mesh visits hundreds of nodes), so "one ray per lane until the whole
// wavefront is done" leaves most lanes idle.  The work is therefore split:
//
//   tlas_kernel      one ray per lane, fully coherent: the 7-slab DOP test against every instance and the
//                    visiting order of intersect_ray (closest hit: increasing entry distance, ties in table
//                    order; any hit: table order from the back).  Writes up to kCandMax (key, slot) pairs per
//                    ray, ended by a PT_NO_NODE slot.
//   traverse_kernel  persistent: every lane is a small state machine and a wavefront keeps pulling rays from
//                    a global cursor:
//                        IDLE   no ray                      \  service: accept the hit of the finished instance,
//                        NEXT   ray needs its next instance /  retire or claim a ray, enter the next candidate
//                        INNER  current reference is a node -> one node step
//                        LEAF   current reference is a leaf -> one triangle test
//                    Per iteration the wavefront runs the phase that most lanes wait for, so its instructions
//                    are shared.  A lane's own sequence of instance visits, node visits and triangle tests is
//                    exactly that of the simple kernel above; only the interleaving across lanes changes.
#ifndef PT_SERVICE_MIN
#define PT_SERVICE_MIN 24
#endif
#ifndef PT_BURST
#define PT_BURST 8
#endif
#ifndef PT_SERVICE_MIN_WIDE
#define PT_SERVICE_MIN_WIDE 32  // measured: 16 / 24 / 32 / 40 / 48 -> extend 26.7 / 26.2 / 25.7 / 26.2 / 27.5 ms (README stand-in, 16 spp)
#endif
#ifndef PT_WIDE_SORT
#define PT_WIDE_SORT 2
#endif
#ifndef PT_WIDE_BURST
#define PT_WIDE_BURST PT_BURST
#endif
#ifdef PT_NO_UNROLL
#define PT_BURST_UNROLL _Pragma("unroll 1")
#else
#define PT_BURST_UNROLL
#endif
// a burst of steps ends when fewer than NUM / DEN of the lanes it started with still hold a node (a leaf)
#ifndef PT_KEEP_NUM
#define PT_KEEP_NUM 1u
#endif
#ifndef PT_KEEP_DEN
#define PT_KEEP_DEN 2u
#endif
#ifndef PT_LEAF_KEEP_NUM
#define PT_LEAF_KEEP_NUM 1u
#endif
#ifndef PT_LEAF_KEEP_DEN
#define PT_LEAF_KEEP_DEN 2u
#endif
#ifndef PT_LEAF_BURST
#define PT_LEAF_BURST PT_BURST
#endif
#ifndef PT_XF_LDS
#define PT_XF_LDS 1
#endif
#ifndef PT_OBJ_LDS
#define PT_OBJ_LDS 32  // object records in the shading kernels' LDS copy (256 B each; only as many as the scene has are copied)
#endif
#ifndef PT_CLAIM_STRADDLE
#define PT_CLAIM_STRADDLE 1
#endif
#ifndef PT_SEVEN_WAVES
#define PT_SEVEN_WAVES 7  // wavefronts per SIMD the second form of the four-wide kernel is compiled for
#endif
#ifndef PT_RAY_BLOCK
#define PT_RAY_BLOCK 128u
#endif
#ifndef PT_RAY_BLOCK_SMALL
#define PT_RAY_BLOCK_SMALL 64u  // launches with fewer than two large blocks per wavefront of the grid
#endif

// Ray claiming of the persistent kernels.  The launch's rays are cut into 8 contiguous segments, one per XCD
// (each XCD has its own L2, and neighbours in the queue are neighbours on the screen, so a segment's working set
// is a part of the scene).  A wavefront claims blocks of PT_RAY_BLOCK rays from the segment of the XCD it runs
// on with ONE atomic on that segment's cursor (a single word sustains well under 100 M atomics/s); when a
// segment is used up it moves on to the next one, so every ray is claimed wherever the wavefronts were placed.
// A launch too small to give every wavefront of its grid two such blocks (a 640 x 360 preview frame, the stragglers at
// the end of a batch) is cut into blocks of PT_RAY_BLOCK_SMALL instead: with 128-ray blocks half of its wavefronts
// would find nothing to claim while the others work through two rounds (preview frame: traversal 239 -> 184 us).
struct RayClaim {
    uint32_t seg, tried;         // current segment, segments found empty so far
    uint32_t next, end;          // rays [next, end) are owned by this wavefront
    bool exhausted;              // all 8 segments are used up (wave-uniform like the rest)
};
constexpr uint32_t kCursorStride = 32;  // cursors sit 128 bytes apart
__device__ __forceinline__ RayClaim make_claim() {
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return RayClaim{xcc & 7u, 0u, 0u, 0u, false};
}
// (The claim is taken and returned by value and its fields are written in one place each: a structure updated through a
// reference from several branches ends up in scratch memory, read back on every service pass.)
__device__ __forceinline__ RayClaim claim_block(RayClaim c, uint32_t* __restrict__ cursors, uint32_t n, uint32_t lane) {
    const uint32_t rb = (n < 2u * kBlock * gridDim.x) ? PT_RAY_BLOCK_SMALL : PT_RAY_BLOCK;
    const uint32_t seg_len = (((n + 7u) / 8u) + rb - 1u) / rb * rb;
    uint32_t seg = c.seg, tried = c.tried, next = 0, end = 0;
    bool exhausted = c.exhausted, found = false;
    while (!exhausted && !found) {
        uint32_t off = 0;
        if (lane == 0) off = atomicAdd(cursors + seg * kCursorStride, rb);
        off = __builtin_amdgcn_readfirstlane(off);
        const uint64_t seg_first = static_cast<uint64_t>(seg) * seg_len;
        const uint64_t seg_end = (seg_first + seg_len < n) ? seg_first + seg_len : n;
        found = off < seg_len && seg_first + off < seg_end;
        if (found) {
            next = static_cast<uint32_t>(seg_first + off);
            end = static_cast<uint32_t>((seg_first + off + rb < seg_end) ? seg_first + off + rb : seg_end);
        } else {
            seg = (seg + 1u) & 7u;
            exhausted = (++tried == 8u);
        }
    }
    return RayClaim{seg, tried, next, end, exhausted};
}

template <bool COUNT>
__global__ __launch_bounds__(kBlock) void tlas_kernel(SceneDev sc, const float4* __restrict__ ray_o,
                                                      const float4* __restrict__ ray_d, uint32_t n, CandQueue cand,
                                                      TraceCounters* counters) {
    const uint32_t gid = blockIdx.x * kBlock + threadIdx.x;
    if (gid >= n) return;
    const float4 ro = ray_o[gid], rd = ray_d[gid];
    const bool any_hit = ((__float_as_uint(rd.w) >> 8) & 1u) != 0;
    const DopRay pr = dop_projections(ld3(ro), ld3(rd));
    float key[kCandMax];
    uint32_t slot[kCandMax];
#pragma unroll
    for (int j = 0; j < kCandMax; ++j) { key[j] = kInfF; slot[j] = PT_NO_NODE; }
    const uint32_t n_inst = sc.n_instances;
    bool finite = true;
#pragma unroll
    for (int i = 0; i < 7; ++i) finite = finite && (fabsf(pr.ip[i]) < kInfF);
    const bool wave_finite = __ballot(!finite) == 0ull;  // (lanes past the end of the queue have left already)
    uint32_t wave_filled = 0;  // wave-uniform: the longest list of any lane, so the entry at wave_filled is empty in every lane
    for (uint32_t k = 0; k < n_inst; ++k) {
        const float dist = wave_finite ? dop_forward_distance_finite(sc.instances[k].dop, pr) : dop_forward_distance(sc.instances[k].dop, pr);
        if (__ballot(dist < kInfF) == 0ull) continue;  // no ray of this wavefront touches the instance: skip the insertion
        bool grew = false;
        if (dist < kInfF) {
            // sorted insertion; strict '<' keeps table order among equal distances
            float ck = any_hit ? -static_cast<float>(k) : dist;
            uint32_t cs = k;
            // The walk covers entries 0 .. wave_filled (uniform: nothing is stored beyond the longest list of the wavefront).
            // One straight-line copy of the walk per length, chosen by a scalar branch: a loop that is left early makes the
            // compiler move the sixteen list registers around at every way out.
            auto walk = [&](auto steps) {
                constexpr int kSteps = decltype(steps)::value;
                // Once the new entry has found its place every later entry moves down by one, whatever its key: asking
                // "is the carried entry smaller" again would leave it ABOVE an entry of equal distance (two instances whose
                // bounds share a plane), the pair would swap places at every insertion in front of them, and when the list
                // is full the one that falls off could be the one with the smaller slot — which the continuation pass,
                // looking for entries after (distance, slot) of the last stored one, never finds.
                bool moving = false;
#pragma unroll
                for (int j = 0; j < kSteps; ++j) {
                    const bool sw = moving || (ck < key[j]) || (slot[j] == PT_NO_NODE);
                    const float tk = key[j];
                    const uint32_t ts = slot[j];
                    key[j] = sw ? ck : tk;
                    slot[j] = sw ? cs : ts;
                    ck = sw ? tk : ck;
                    cs = sw ? ts : cs;
                    moving = sw;
                }
                // only a lane whose list was the longest fills the entry at wave_filled (lanes that sit this instance out
                // hold fewer entries than that already)
                grew = __ballot(slot[kSteps - 1] != PT_NO_NODE) != 0ull;
            };
            static_assert(kCandMax == 8, "one case per list length");
            switch (wave_filled) {
                case 0: walk(std::integral_constant<int, 1>{}); break;
                case 1: walk(std::integral_constant<int, 2>{}); break;
                case 2: walk(std::integral_constant<int, 3>{}); break;
                case 3: walk(std::integral_constant<int, 4>{}); break;
                case 4: walk(std::integral_constant<int, 5>{}); break;
                case 5: walk(std::integral_constant<int, 6>{}); break;
                case 6: walk(std::integral_constant<int, 7>{}); break;
                default: walk(std::integral_constant<int, 8>{}); grew = false; break;  // (a full list cannot grow)
            }
        }
        if (__ballot(grew) != 0ull) ++wave_filled;
    }
    // An entry's slot word carries kCandLast when the stored list ends with it, so a list costs no terminator entry and
    // the traversal kernel knows without a read that nothing follows; an empty list is one entry with the slot PT_NO_NODE.
#pragma unroll
    for (int j = 0; j < kCandMax; ++j)
        if (j == 0 || slot[j] != PT_NO_NODE) {
            const bool last = (j + 1 < kCandMax) && (slot[j] != PT_NO_NODE) && (slot[j + 1 < kCandMax ? j + 1 : j] == PT_NO_NODE);
            const uint32_t word = slot[j] | (last ? kCandLast : 0u);
            if (j == 0) cand.h1[gid] = 0u;  // "no hit" until the traversal kernel accepts one: a miss then writes nothing
            if (j == 0) cand.first()[gid] = word;  // (the first candidate's distance is never looked at)
            else cand.e[static_cast<size_t>(j) * cand.stride + gid] = make_uint2(__float_as_uint(any_hit ? 0.0f : key[j]), word);
        }
    if (COUNT) atomicAdd(&counters->inst_tested, static_cast<unsigned long long>(n_inst));
    if (COUNT && slot[0] == PT_NO_NODE) atomicAdd(&counters->census[1][1], 1ull);  // rays that touch no instance (an empty list)
}

// The same pass for scenes with many instances.  The reference tests every object's bound for every ray
// (acceleration_structure.cc:599-607), which is what tlas_kernel does while that is cheap; here a binary tree over the
// bounds' axis-aligned boxes (the first three slabs of a 14-DOP are its box) culls first, and only the instances whose
// box the ray enters get the 7-slab test.  The tree is stored depth first and threaded (TlasNode), so a ray walks it
// without a stack, in one fixed order.  A box test uses the arithmetic of those three slabs, which is monotonic in the
// plane positions: a ray that enters an instance's box enters every box around it, so no candidate is lost.  The
// candidates come in tree order here, so the insertion orders by (distance, table index) explicitly.
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void tlas_tree_kernel(SceneDev sc, const float4* __restrict__ ray_o,
                                                           const float4* __restrict__ ray_d, uint32_t n, CandQueue cand,
                                                           TraceCounters* counters) {
    __shared__ float4 s_nodes[kTlasLds * 2];  // the first nodes of the tree (depth-first order: its top and its left flank)
    const uint32_t tid = threadIdx.x;
    const uint32_t gid = blockIdx.x * kBlock + tid;
    const uint32_t n_nodes = sc.n_tlas;
    const uint32_t n_lds = n_nodes < kTlasLds ? n_nodes : kTlasLds;
    for (uint32_t i = tid; i < n_lds * 2u; i += kBlock) s_nodes[i] = reinterpret_cast<const float4*>(sc.tlas)[i];
    __syncthreads();
    if (gid >= n) return;
    const float4 ro = ray_o[gid], rd = ray_d[gid];
    const bool any_hit = ((__float_as_uint(rd.w) >> 8) & 1u) != 0;
    const DopRay pr = dop_projections(ld3(ro), ld3(rd));
    const bool finite3 = (fabsf(pr.ip[0]) < kInfF) && (fabsf(pr.ip[1]) < kInfF) && (fabsf(pr.ip[2]) < kInfF);
    float key[kCandMax];
    uint32_t slot[kCandMax];
#pragma unroll
    for (int j = 0; j < kCandMax; ++j) { key[j] = kInfF; slot[j] = PT_NO_NODE; }
    uint32_t tested = 0;
    for (uint32_t i = 0; i < n_nodes;) {
        const float4* np = (i < n_lds) ? s_nodes + i * 2u : reinterpret_cast<const float4*>(sc.tlas + i);
        const float4 a = np[0], b = np[1];  // lo.xyz skip | hi.xyz inst
        bool inside = true;
        if (finite3) {  // (a zero direction component can make a slab NaN: such rays skip the culling)
            const float t0x = (a.x - pr.op[0]) * pr.ip[0], t1x = (b.x - pr.op[0]) * pr.ip[0];
            const float t0y = (a.y - pr.op[1]) * pr.ip[1], t1y = (b.y - pr.op[1]) * pr.ip[1];
            const float t0z = (a.z - pr.op[2]) * pr.ip[2], t1z = (b.z - pr.op[2]) * pr.ip[2];
            const float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)), __builtin_fminf(t0z, t1z));
            const float tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fmaxf(t0y, t1y)), __builtin_fmaxf(t0z, t1z));
            inside = (tmin <= tmax) && (tmax > 0.0f);
        }
        const uint32_t skip = __float_as_uint(a.w), inst = __float_as_uint(b.w);
        if (!inside) {
            i = skip;
            continue;
        }
        if (inst == PT_NO_NODE) {
            ++i;  // an inner node: its first child is the next node
            continue;
        }
        i = skip;
        const uint32_t k = inst;
        ++tested;
        // (the per-lane form of the 7-slab test: no wave-level early out, the lanes are at different instances)
        const float* dop = sc.instances[k].dop;
        float tmin = -3.402823466e+38f, tmax = 3.402823466e+38f;
        bool hit = true;
#pragma unroll
        for (int s7 = 0; s7 < 7; ++s7) {
            float t0 = (dop[2 * s7] - pr.op[s7]) * pr.ip[s7];
            float t1 = (dop[2 * s7 + 1] - pr.op[s7]) * pr.ip[s7];
            if (t0 > t1) { float sw = t0; t0 = t1; t1 = sw; }
            tmin = (tmin < t0) ? t0 : tmin;
            tmax = (t1 < tmax) ? t1 : tmax;
            if (hit && (tmax < tmin)) hit = false;
        }
        const float dist = forward_distance(hit, tmin, tmax);
        if (dist < kInfF) {
            float ck = any_hit ? -static_cast<float>(k) : dist;
            uint32_t cs = k;
#pragma unroll
            for (int j = 0; j < kCandMax; ++j) {
                const bool sw = (slot[j] == PT_NO_NODE) || (ck < key[j]) || (ck == key[j] && cs < slot[j]);
                const float tk = key[j];
                const uint32_t ts = slot[j];
                key[j] = sw ? ck : tk;
                slot[j] = sw ? cs : ts;
                ck = sw ? tk : ck;
                cs = sw ? ts : cs;
            }
        }
    }
    // An entry's slot word carries kCandLast when the stored list ends with it, so a list costs no terminator entry and
    // the traversal kernel knows without a read that nothing follows; an empty list is one entry with the slot PT_NO_NODE.
#pragma unroll
    for (int j = 0; j < kCandMax; ++j)
        if (j == 0 || slot[j] != PT_NO_NODE) {
            const bool last = (j + 1 < kCandMax) && (slot[j] != PT_NO_NODE) && (slot[j + 1 < kCandMax ? j + 1 : j] == PT_NO_NODE);
            const uint32_t word = slot[j] | (last ? kCandLast : 0u);
            if (j == 0) cand.h1[gid] = 0u;  // "no hit" until the traversal kernel accepts one: a miss then writes nothing
            if (j == 0) cand.first()[gid] = word;  // (the first candidate's distance is never looked at)
            else cand.e[static_cast<size_t>(j) * cand.stride + gid] = make_uint2(__float_as_uint(any_hit ? 0.0f : key[j]), word);
        }
    if (COUNT) atomicAdd(&counters->inst_tested, static_cast<unsigned long long>(tested));
}

// Continuation of the visiting order for rays that consumed kCandMax candidates: rescan all DOPs.
__device__ __noinline__ uint2 tlas_rescan(const SceneDev& sc, f3 ow, f3 dw, bool any_hit, float last_key, uint32_t last_slot) {
    const DopRay pr = dop_projections(ow, dw);
    uint32_t next = PT_NO_NODE;
    float best_d = kInfF;
    if (any_hit) {
        for (uint32_t k = last_slot; k-- > 0;)
            if (dop_forward_distance(sc.instances[k].dop, pr) < kInfF) { next = k; best_d = 0.0f; break; }
    } else {
        for (uint32_t k = 0; k < sc.n_instances; ++k) {
            const float dist = dop_forward_distance(sc.instances[k].dop, pr);
            const bool after = (dist > last_key) || (dist == last_key && k > last_slot);
            if (after && dist < best_d) { best_d = dist; next = k; }
        }
    }
    return make_uint2(__float_as_uint(best_d), next);
}

// BBox::ray_volume_intersection + forward_hit_distance for lanes whose inverse direction is finite: no
// operand can be NaN then, so hardware min/max/min3/max3 return what the reference's compare-and-select
// forms return (up to the sign of a zero, which no later comparison observes).
__device__ __forceinline__ float box_entry_fast(float lox, float loy, float loz, float hix, float hiy, float hiz, f3 o, f3 inv) {
    const float ax = (lox - o.x) * inv.x, ay = (loy - o.y) * inv.y, az = (loz - o.z) * inv.z;
    const float bx = (hix - o.x) * inv.x, by = (hiy - o.y) * inv.y, bz = (hiz - o.z) * inv.z;
    const float t0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
    const float t1 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
    return ((t0 <= t1) && (t1 > 0.0f)) ? __builtin_fmaxf(t0, 0.0f) : kInfF;
}

// Wavefront priority around the loads of a traversal step: a wavefront that is about to ask for its node or its triangles
// goes ahead of wavefronts that are computing, so that its round trip starts earlier (traversal -0.6 %; the same around the
// loads of the service phase or of the shading kernel measured without gain).
#ifndef PT_PRIO
#define PT_PRIO 1
#endif
#if PT_PRIO
#define PT_PRIO_UP __builtin_amdgcn_s_setprio(PT_PRIO)
#define PT_PRIO_DOWN __builtin_amdgcn_s_setprio(0)
#else
#define PT_PRIO_UP
#define PT_PRIO_DOWN
#endif
enum : uint32_t { F_ANY = 1u, F_CULL = 2u, F_SLOW = 4u, F_BHIT = 8u, F_LAST = 16u };  // F_LAST: the candidate list has no further entry
// SEVEN: the LDS budget of seven workgroups per CU instead of six (a shorter LDS stack, fewer cached nodes and transforms),
// for the kernel that is compiled for seven wavefronts per SIMD.
// XF_LDS: every instance transform of the scene is in the kernel's LDS copy, so a transform is read with LDS instructions
// (no work for the vector memory path's address unit, and its waits do not cover the loads from memory in flight).
// XF_MANY (seven-wave form only): twice the transforms in LDS, paid for with fewer cached nodes.
// LEAF64 (four-wide forms): leaves are aligned 64-byte records (Leaf64) instead of runs of 48-byte triangles.
template <bool COUNT, bool WIDE, bool SEVEN, bool XF_LDS, bool XF_MANY = false, bool LEAF64 = false>
__device__ __forceinline__ void traverse_body(const SceneDev& sc, const float4* __restrict__ ray_o,
                                              const float4* __restrict__ ray_d, uint32_t n, CandQueue cand,
                                              float4* __restrict__ hit0, uint32_t* __restrict__ hit1,
                                              uint32_t* __restrict__ spill, uint32_t spill_depth, uint32_t spill_stride,
                                              uint32_t* __restrict__ next_ray, uint32_t* __restrict__ overflow,
                                              TraceCounters* counters) {
    constexpr int kStack = SEVEN ? kStackLds7 : kStackLds;
    static_assert(!XF_MANY || (SEVEN && XF_LDS), "the larger transform table is a form of the seven-wave kernel");
    constexpr int kTop = SEVEN ? (XF_MANY ? kTopNodes7Many : kTopNodes7) : kTopNodes;
    constexpr uint32_t kXf = SEVEN ? (XF_MANY ? 2u * kXformLds7 : kXformLds7) : kXformLds;
    __shared__ uint32_t s_stack[kStack][kBlock];
    // The most visited nodes (the first of the node array), 64 bytes each, node after node.  Layouts that spread the
    // lanes' reads over more banks (quarter-major, 80-byte stride) measured 20-40 % slower for the whole kernel.
    __shared__ float4 s_top[kTop * 4];
    // the transforms of the first instances (128 B each): read once per instance a ray enters, and again per accepted hit
    __shared__ float4 s_xf[kXf * (sizeof(DevXform) / 16)];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t gthread = blockIdx.x * kBlock + tid;  // owner of one spill column
    TraceCount cnt{};
    static_assert(kTopNodes4 <= kTopNodes && sizeof(Node4) == sizeof(NodeQ), "the four-wide node cache fits the same LDS array");
    static_assert(!SEVEN || WIDE, "the seven-wave budget is for the four-wide kernel");
    static_assert(!LEAF64 || WIDE, "leaf records belong to the four-wide kernels");
    // Which phase runs when lanes wait at nodes and at leaves: node steps while (lanes at nodes) >= (lanes at leaves) * leaf_vote.
    // With a weight of 4 a leaf step runs for as few as a fifth of the traversing lanes: the lanes it releases go back to the node
    // steps, which are most of the work of a deep closest-hit traversal (readme frame: traversal -1.5 %, binary nodes -2.2 %,
    // 10 M-triangle frame -1.9 %); short traversals and any-hit rays, which end at their leaf, run 1-2 % slower that way and keep 1.
    const uint32_t leaf_vote = sc.leaf_vote;
    const uint32_t n_top_scene = WIDE ? sc.n_top_nodes4 : sc.n_top_nodes;
#ifdef PT_NO_TOP_NODES  // experiment: no node is served from LDS (the node loads are then plain global loads)
    const uint32_t n_top = 0u * n_top_scene;
#else
    const uint32_t n_top = n_top_scene < static_cast<uint32_t>(kTop) ? n_top_scene : static_cast<uint32_t>(kTop);
#endif
    {
        const float4* src = WIDE ? reinterpret_cast<const float4*>(sc.nodes4) : reinterpret_cast<const float4*>(sc.nodes);
        for (uint32_t i = tid; i < n_top * 4u; i += kBlock) s_top[i] = src[i];
        const uint32_t n_xf = (sc.n_instances < kXf ? sc.n_instances : kXf) * (sizeof(DevXform) / 16);
        if (tid < n_xf) s_xf[tid] = reinterpret_cast<const float4*>(sc.xforms)[tid];
    }
    __syncthreads();
    auto xform_record = [&](uint32_t k) -> const DevXform& {
        if (XF_LDS) return reinterpret_cast<const DevXform*>(s_xf)[k < kXf ? k : 0u];
        return (k < kXf) ? reinterpret_cast<const DevXform*>(s_xf)[k] : sc.xforms[k];
    };

    // ---- per-lane state ----
    // `cur` doubles as the lane's state: PT_IDLE = no ray, PT_NO_NODE = the ray needs its next instance,
    // bit 31 clear = node to visit, bit 31 set (below PT_IDLE) = triangle to test next.
    uint32_t cur = PT_IDLE;
    uint32_t rid = 0;            // ray index
    uint32_t cursor = 0;         // next candidate to read
    uint32_t flags = 0;          // F_*
    float b_t = kInfF;           // world distance of the best accepted hit (its record is already in hit0/hit1)
    uint32_t slot = PT_NO_NODE;  // instance being traversed
    f3 o = mk3(0.0f), d = mk3(0.0f), inv = mk3(0.0f);
    float h_t = kInfF, h_A = 0.0f, h_B = 0.0f;
    uint32_t h_tri = 0;
    bool h_hit = false, h_back = false;
    int sp = 0;
    RayClaim claim = make_claim();

    auto is_node = [](uint32_t c) { return static_cast<int32_t>(c) >= 0; };
    auto is_leaf = [](uint32_t c) { return static_cast<int32_t>(c) < -2; };
    // The top of the stack is read when a step starts, together with the node or triangle, so that a pop costs
    // no second round trip; only the rare entry beyond the LDS part is fetched when it is needed.
    auto peek = [&]() -> uint32_t {
        const int at = (sp > kStack) ? kStack - 1 : sp - 1;
        return (sp > 0) ? s_stack[at][tid] : PT_NO_NODE;
    };
    auto pop_to = [&](uint32_t top) {
        if (sp > kStack) top = spill[static_cast<size_t>(sp - 1 - kStack) * spill_stride + gthread];
        cur = top;
        sp = (sp > 0) ? sp - 1 : 0;
    };

    // One step over a four-wide node (its four quarters q0 q1 q2 rr already loaded): enter the nearest child whose box is
    // entered before the current hit, push the others far to near, or pop.
    auto wide_node_step = [&](float4 q0, float4 q1, float4 q2, float4 rr, uint32_t top) {
            // planes: origin + scale * q.  With the ray folded in, a plane's distance is A + q * B per axis.
            const uint32_t lqx = __float_as_uint(q1.z), lqy = __float_as_uint(q1.w), lqz = __float_as_uint(q2.x);
            const uint32_t hqx = __float_as_uint(q2.y), hqy = __float_as_uint(q2.z), hqz = __float_as_uint(q2.w);
            float d0, d1, d2, d3;
            if (!(flags & F_SLOW)) {
                const float ax = (q0.x - o.x) * inv.x, ay = (q0.y - o.y) * inv.y, az = (q0.z - o.z) * inv.z;
                const float bx = q0.w * inv.x, by = q1.x * inv.y, bz = q1.y * inv.z;
                // Which of a slab's two planes is entered first depends on the sign of the direction only (a plane's
                // distance A + q * B grows with q when B >= 0 and shrinks otherwise; the step is never negative, so B
                // has the sign of the inverse direction): choosing the near and far index words of all four children
                // once per axis replaces a min and a max per child and axis — the same two values, fewer instructions.
                const bool neg_x = inv.x < 0.0f, neg_y = inv.y < 0.0f, neg_z = inv.z < 0.0f;
                const uint32_t nqx = neg_x ? hqx : lqx, fqx = neg_x ? lqx : hqx;
                const uint32_t nqy = neg_y ? hqy : lqy, fqy = neg_y ? lqy : hqy;
                const uint32_t nqz = neg_z ? hqz : lqz, fqz = neg_z ? lqz : hqz;
                auto entry = [&](int k) {
                    const int sh = 8 * k;
                    const float nx = __builtin_fmaf(static_cast<float>((nqx >> sh) & 0xffu), bx, ax);
                    const float fx = __builtin_fmaf(static_cast<float>((fqx >> sh) & 0xffu), bx, ax);
                    const float ny = __builtin_fmaf(static_cast<float>((nqy >> sh) & 0xffu), by, ay);
                    const float fy = __builtin_fmaf(static_cast<float>((fqy >> sh) & 0xffu), by, ay);
                    const float nz = __builtin_fmaf(static_cast<float>((nqz >> sh) & 0xffu), bz, az);
                    const float fz = __builtin_fmaf(static_cast<float>((fqz >> sh) & 0xffu), bz, az);
                    // The interval of the box is cut to [0, current hit distance] before it is tested for being empty: one
                    // comparison decides "entered, in front, and nearer than the hit".  (A box entered exactly at the hit
                    // distance, or touched exactly at the origin, counts as entered here: a visit more, never a hit less.)
                    const float t0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(nx, ny), nz), 0.0f);
                    const float t1 = __builtin_fminf(__builtin_fminf(__builtin_fminf(fx, fy), fz), h_t);
                    return (t0 <= t1) ? t0 : kInfF;
                };
                d0 = entry(0); d1 = entry(1); d2 = entry(2); d3 = entry(3);
            } else {  // an infinite inverse direction: planes decoded, exact select forms
                auto entry = [&](int k) {
                    const int sh = 8 * k;
                    // (an unused slot holds an inverted box; this test is symmetric in a slab's two planes and would take
                    // it for a box that spans the node, so such a slot is skipped here explicitly)
                    if (((lqx >> sh) & 0xffu) > ((hqx >> sh) & 0xffu)) return kInfF;
                    bool hh;
                    float dd;
                    box_test(q0.x + q0.w * static_cast<float>((lqx >> sh) & 0xffu), q0.y + q1.x * static_cast<float>((lqy >> sh) & 0xffu),
                             q0.z + q1.y * static_cast<float>((lqz >> sh) & 0xffu), q0.x + q0.w * static_cast<float>((hqx >> sh) & 0xffu),
                             q0.y + q1.x * static_cast<float>((hqy >> sh) & 0xffu), q0.z + q1.y * static_cast<float>((hqz >> sh) & 0xffu),
                             o, inv, hh, dd);
                    return (h_t > dd) ? dd : kInfF;  // a candidate when its box is entered nearer than the current hit
                };
                d0 = entry(0); d1 = entry(1); d2 = entry(2); d3 = entry(3);
            }
            // (a node with fewer than four children has its unused slots filled with an inverted box around a leaf that holds
            // one degenerate triangle, so the fast path needs no "exists" test: see collapse_to_wide in pt_scene_build.cc)
            uint32_t r0 = __float_as_uint(rr.x), r1 = __float_as_uint(rr.y), r2 = __float_as_uint(rr.z), r3 = __float_as_uint(rr.w);
            auto order = [](float& da, uint32_t& ra, float& db, uint32_t& rb) {
                const bool swap = db < da;
                const float dn = swap ? db : da, df = swap ? da : db;
                const uint32_t rn = swap ? rb : ra, rf = swap ? ra : rb;
                da = dn; db = df; ra = rn; rb = rf;
            };
#if PT_WIDE_SORT == 2
            order(d0, r0, d1, r1);
            order(d2, r2, d3, r3);
            order(d0, r0, d2, r2);
            order(d1, r1, d3, r3);
            order(d1, r1, d2, r2);
#else  // the nearest child to the front, the others in slot order
            order(d0, r0, d1, r1);
            order(d2, r2, d3, r3);
            order(d0, r0, d2, r2);
#endif
            auto push = [&](uint32_t ref) {
                if (sp < kStack) s_stack[sp][tid] = ref;
                else if (static_cast<uint32_t>(sp - kStack) < spill_depth)
                    spill[static_cast<size_t>(sp - kStack) * spill_stride + gthread] = ref;
                ++sp;
            };
            if (d0 < kInfF) {  // nearest first; the others wait on the stack, the nearer the higher
                if (d3 < kInfF) push(r3);
                if (d2 < kInfF) push(r2);
                if (d1 < kInfF) push(r1);
                cur = r0;
            } else {
                pop_to(top);
            }
        
    };

    for (;;) {
        // ---- census of lane states (wave-uniform scalars) ----
        const unsigned long long m_idle = __ballot(cur == PT_IDLE);
        const uint32_t n_idle = __popcll(m_idle);
        const uint32_t n_next = __popcll(__ballot(cur == PT_NO_NODE));
        const uint32_t n_inner = __popcll(__ballot(is_node(cur)));
        const uint32_t n_leaf = 64u - n_idle - n_next - n_inner;

        auto tally = [&](int phase, int steps) {
            if (lane == 0) {
                cnt.census[phase][0] += 1;
                cnt.census[phase][1] += steps;
                cnt.census[phase][2] += n_idle * steps;
                cnt.census[phase][3] += n_next * steps;
                cnt.census[phase][4] += n_inner * steps;
                cnt.census[phase][5] += n_leaf * steps;
            }
        };
        const uint32_t n_wait = n_next + (claim.exhausted ? 0u : n_idle);  // lanes that a service pass can move on
        if (n_wait && (n_wait >= (WIDE ? PT_SERVICE_MIN_WIDE : PT_SERVICE_MIN) || (n_inner + n_leaf) == 0u)) {
            // ================= service: finish instances, hand out rays, enter instances =================
            // One pass serves every lane that is not traversing, so a lane waits once between two instances
            // instead of once per kind of service:
            //   (1) lanes whose instance is finished accept its hit and pick their next candidate, or retire;
            //   (2) idle lanes, the just retired ones included, claim new rays and pick their first candidate;
            //   (3) all lanes that picked a candidate transform their ray into its object space together.
            if (COUNT) tally(0, 1);
            uint32_t enter_slot = PT_NO_NODE;
            f3 ow = mk3(0.0f), dw = mk3(0.0f);
            auto pick = [&](bool fresh) {
                // A ray that has consumed the last entry of its candidate list needs neither its direction nor another
                // list entry, and its origin only to turn a hit's distance into world space.
                const bool more = fresh || !(flags & F_LAST);
                // The next list entry is asked for together with the ray, both words at once, although whether it is needed
                // (an any-hit ray that just found its hit needs none) and whether its distance is looked at is only known
                // after the ray has arrived: one round trip for the whole pick instead of up to three in a row, while every
                // lane of the wavefront waits.  The three loads are unconditional so that nothing separates them; a lane
                // that needs none of a record reads record 0 instead (one cache line for all such lanes, no bytes from memory).
                const bool entry_stored = more && cursor < kCandMax;
                const float4 ro = ray_o[(more || h_hit) ? rid : 0u];
                const float4 rd = ray_d[more ? rid : 0u];
                // (the first candidate is a slot word of its own; a ray that has no hit yet enters it whatever its distance)
                const bool first_entry = cursor == 0u;
                const uint32_t e0 = cand.first()[(entry_stored && first_entry) ? rid : 0u];
                const uint2 e = cand.e[(entry_stored && !first_entry) ? static_cast<size_t>(cursor) * cand.stride + rid : 0u];
                uint32_t e_x = first_entry ? 0u : e.x, e_y = first_entry ? e0 : e.y, rd_w = __float_as_uint(rd.w);
                // (an empty statement that "uses" the entry and the ray's flag word here: without it the compiler moves each
                // word's load down to the branch that looks at it, behind the wait for the ray, and the round trips are in
                // a row again)
                asm volatile("" : "+v"(e_x), "+v"(e_y), "+v"(rd_w));
                const float e_key = __uint_as_float(e_x);
                const uint32_t e_slot = e_y;
                ow = ld3(ro);
                dw = ld3(rd);
                if (fresh) {
                    if ((rd_w >> 8) & 1u) flags |= F_ANY;
                } else if (h_hit) {
                    // back to a world-space distance and acceptance (acceleration_structure.cc:704-706, 616-623)
                    const DevXform& xf = xform_record(slot);
                    uint32_t object_index = xf.object_index;
                    const f3 p = o + d * h_t;
                    const f3 p_ws = xform(xf.model, p, 1.0f);
                    // (read with the matrix and kept there: read after the comparison below, the store of the flag word would
                    // wait for this load and with it for the store of the distance record before it)
                    asm volatile("" : "+v"(object_index));
                    const float t_ws = length3(p_ws - ow);
                    if ((t_ws > 0.0f) && (t_ws < b_t)) {
                        b_t = t_ws;
                        flags |= F_BHIT;
                        hit0[rid] = make_float4(t_ws, h_A, h_B, __uint_as_float(h_tri));
                        hit1[rid] = object_index | (h_back ? 0x40000000u : 0u) | 0x80000000u;
                    }
                }
                const bool any_hit = (flags & F_ANY) != 0;
                bool deferred = false;
                if (more && !(any_hit && (flags & F_BHIT))) {
                    if (entry_stored) {
                        ++cursor;
                        const bool nearer = e_key < b_t;
                        if (e_slot != PT_NO_NODE) {
                            if (e_slot & kCandLast) flags |= F_LAST;
                            if (any_hit || nearer) enter_slot = e_slot & ~kCandLast;
                        }
                    } else {
                        deferred = true;  // all stored candidates are used up: the continuation pass takes the ray
                    }
                }
                if (enter_slot == PT_NO_NODE) {  // the ray is done here
                    // a miss writes nothing: the instance pass has set every flag word of the launch to "no hit" (one streamed
                    // word per ray instead of a scattered one that leaves the L2 as a partial sector), and nothing reads the
                    // distance / barycentrics record of a ray whose flag word says so
                    if (deferred) overflow[1u + atomicAdd(overflow, 1u)] = rid;
                    cur = PT_IDLE;
                }
            };
            if (cur == PT_NO_NODE) pick(false);
            bool fresh = false;
            if (!claim.exhausted) {
                const unsigned long long m_free = __ballot(cur == PT_IDLE);
                const uint32_t n_free = __popcll(m_free);
                const uint32_t rank = __popcll(m_free & lt_mask);
                // What is left of the wavefront's block goes to the first free lanes; when that is not enough for all of
                // them, the next block is claimed in the same pass and the other lanes start there — a lane left without a
                // ray here would sit out every step until the next service pass.
                const uint32_t rest_first = claim.next;
                const uint32_t rest = claim.end - claim.next;
                const bool claims = PT_CLAIM_STRADDLE ? (rest < n_free) : (n_free && rest == 0u);
                if (claims) claim = claim_block(claim, next_ray, n, lane);
                const uint32_t from_rest = claims ? rest : 0u;  // lanes served by what was left of the old block
                const uint32_t avail = claim.end - claim.next;                            // rays left in the current block
                uint32_t take = 0xffffffffu;
                if (cur == PT_IDLE) {
                    if (rank < from_rest) take = rest_first + rank;
                    else if (rank - from_rest < avail) take = claim.next + (rank - from_rest);
                }
                if (take != 0xffffffffu) {
                    rid = take;
                    cursor = 0;
                    flags = 0;
                    b_t = kInfF;
                    h_hit = false;
                    fresh = true;
                }
                const uint32_t wanted = n_free - from_rest;  // (from_rest < n_free whenever it is not zero)
                claim.next += (wanted < avail) ? wanted : avail;
            }
            if (fresh) pick(true);
            if (enter_slot != PT_NO_NODE) {
                if (COUNT) ++cnt.inst_entered;
                slot = enter_slot;
                const DevXform& xf = xform_record(slot);
                o = xform(xf.inv, ow, 1.0f);
                d = normalize3(xform(xf.inv, dw, 0.0f));
                inv = f3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
                flags &= ~(F_CULL | F_SLOW);
                if (xf.cull) flags |= F_CULL;
                const bool finite = (fabsf(inv.x) < kInfF) && (fabsf(inv.y) < kInfF) && (fabsf(inv.z) < kInfF);
                if (!finite) flags |= F_SLOW;
                h_t = kInfF; h_A = 0.0f; h_B = 0.0f; h_tri = 0; h_hit = false; h_back = false;
                sp = 0;
                cur = WIDE ? xf.root4 : xf.root;
            }
        } else if (n_inner && n_inner >= n_leaf * leaf_vote) {
            // ================= inner-node steps for the lanes that hold a node =================
            PT_BURST_UNROLL
            for (int burst = 0;;) {
                if (WIDE && is_node(cur)) {
                    if (COUNT) { ++cnt.nodes; if (first_active_lane()) ++cnt.wave_inner; }
                    // The two sources are read by their own instructions (LDS reads for the cached top nodes, global loads
                    // for the rest): one pointer that may point to either makes the compiler emit generic loads, which send
                    // the LDS-served visits through the vector memory path's address unit as well.
                    // One pointer that is either LDS or global: the compiler reads through it with generic loads, which
                    // keeps LDS-served and memory-served lanes in the same four instructions.  Reading the LDS copy with LDS
                    // instructions of its own was measured and is slower (12.2 instead of 11.2 ms per 73.6 M rays): the two
                    // kinds of reads then wait for each other instead of overlapping.
                    const float4* np;
                    PT_PRIO_UP;
                    if (cur < n_top) {
                        if (COUNT) ++cnt.inst_tested;
                        np = s_top + cur * 4u;
                    } else {
                        np = reinterpret_cast<const float4*>(sc.nodes4 + cur);
                    }
                    const float4 q0 = np[0], q1 = np[1], q2 = np[2], rr = np[3];
                    const uint32_t top = peek();
                    PT_PRIO_DOWN;
                    wide_node_step(q0, q1, q2, rr, top);
                }
                if (!WIDE && is_node(cur)) {
                    if (COUNT) { ++cnt.nodes; if (first_active_lane()) ++cnt.wave_inner; }
                    float4 nx, ny, nz, nr;
                    if (cur < n_top) {  // from LDS: no work for the L1's address unit
                        if (COUNT) ++cnt.inst_tested;  // (this kernel tests no instance bounds: counts LDS-served node visits)
                        const float4* np = s_top + cur * 4u;
                        nx = np[0]; ny = np[1]; nz = np[2]; nr = np[3];
                    } else {
                        const float4* np = reinterpret_cast<const float4*>(sc.nodes + cur);
                        nx = np[0]; ny = np[1]; nz = np[2]; nr = np[3];
                    }
                    const uint32_t top = peek();
                    float dl = box_entry_fast(nx.x, ny.x, nz.x, nx.z, ny.z, nz.z, o, inv);
                    float dr = box_entry_fast(nx.y, ny.y, nz.y, nx.w, ny.w, nz.w, o, inv);
                    if (flags & F_SLOW) {  // an infinite inverse direction can produce NaNs: exact select forms
                        bool hl, hr;
                        box_test(nx.x, ny.x, nz.x, nx.z, ny.z, nz.z, o, inv, hl, dl);
                        box_test(nx.y, ny.y, nz.y, nx.w, ny.w, nz.w, o, inv, hr, dr);
                    }
                    const uint32_t left = __float_as_uint(nr.x), right = __float_as_uint(nr.y);
                    // A child is visited when its box is hit in front and nearer than the current hit (a missed box
                    // has distance +inf); the reference pushes the farther child first and pops the nearer one next
                    // — the nearer child stays in a register here.  The distances are never NaN, so the farther
                    // child can only be taken when the nearer one is.
                    const bool l_first = dl < dr;
                    const uint32_t near_ref = l_first ? left : right, far_ref = l_first ? right : left;
                    const float d_near = l_first ? dl : dr, d_far = l_first ? dr : dl;
                    if (h_t > d_far) {
                        if (sp < kStack) s_stack[sp][tid] = far_ref;
                        else if (static_cast<uint32_t>(sp - kStack) < spill_depth)
                            spill[static_cast<size_t>(sp - kStack) * spill_stride + gthread] = far_ref;
                        ++sp;
                        cur = near_ref;
                    } else if (h_t > d_near) {
                        cur = near_ref;
                    } else {
                        pop_to(top);
                    }
                }
                if (++burst >= (WIDE ? PT_WIDE_BURST : PT_BURST)) { if (COUNT) tally(2, burst); break; }
                const uint32_t still = __popcll(__ballot(is_node(cur)));
                if (still * PT_KEEP_DEN < n_inner * PT_KEEP_NUM) { if (COUNT) tally(2, burst); break; }
            }
        } else if (n_leaf) {
            // ================= triangle tests for the lanes that hold a leaf =================
            PT_BURST_UNROLL
            for (int burst = 0;;) {
                if (WIDE && LEAF64 && is_leaf(cur)) {
                    // One aligned 64-byte record a step: the leaf's one or two triangles as their distinct corners.  The edges
                    // p2 - p1 and p3 - p1 are formed here, by the same two subtractions the 48-byte records were made with.
                    if (COUNT) { ++cnt.tris; ++cnt.leaf_records; if (first_active_lane()) ++cnt.wave_tri; }
                    PT_PRIO_UP;
                    const float4* lp = reinterpret_cast<const float4*>(sc.leaves64 + (cur & ~PT_LEAF_BIT));
                    const float4 a = lp[0], b = lp[1], c = lp[2], m = lp[3];
                    const uint32_t top = peek();
                    PT_PRIO_DOWN;
                    const f3 v0 = f3{a.x, a.y, a.z}, v1 = f3{a.w, b.x, b.y}, v2 = f3{b.z, b.w, c.x}, v3 = f3{c.y, c.z, c.w};
                    const uint32_t info = __float_as_uint(m.z);
                    float A, B, t;
                    bool bf;
                    const bool ok = ray_triangle(o, d, v0, v1 - v0, v2 - v0, (flags & F_CULL) != 0, A, B, t, bf);
                    bool stop = false;
                    if (ok && !(t < 0.0f) && !(t >= h_t)) {
                        h_A = A; h_B = B; h_t = t; h_hit = true; h_back = bf;
                        h_tri = __float_as_uint(m.x);
                        stop = (flags & F_ANY) != 0;
                    }
                    if (!stop && (info & 1u)) {
                        if (COUNT) ++cnt.tris;
                        auto corner = [&](uint32_t sel) {  // v0..v3 by a 2-bit selector
                            const bool hi = (sel & 2u) != 0, odd = (sel & 1u) != 0;
                            const f3 lo_pair = odd ? v1 : v0, hi_pair = odd ? v3 : v2;
                            return hi ? hi_pair : lo_pair;
                        };
                        const f3 p1 = corner(info >> 8), p2 = corner(info >> 10), p3 = corner(info >> 12);
                        const bool ok2 = ray_triangle(o, d, p1, p2 - p1, p3 - p1, (flags & F_CULL) != 0, A, B, t, bf);
                        if (ok2 && !(t < 0.0f) && !(t >= h_t)) {
                            h_A = A; h_B = B; h_t = t; h_hit = true; h_back = bf;
                            h_tri = __float_as_uint(m.y);
                            stop = (flags & F_ANY) != 0;
                        }
                    }
                    if (stop) {
                        cur = PT_NO_NODE;
                    } else if (info & 2u) {
                        cur = cur + 1u;
                    } else {
                        pop_to(top);
                    }
                }
                if (WIDE && !LEAF64 && is_leaf(cur)) {
                    // Two triangles a step: the trees of this layout have leaves of at most two, and fetching both at once
                    // takes one of the two dependent round trips out of a leaf visit.  (The record after the last
                    // triangle of the array is padding.)  Tested in leaf order, so the first found still wins a tie.
                    if (COUNT) { ++cnt.tris; if (first_active_lane()) ++cnt.wave_tri; }
                    PT_PRIO_UP;
                    const float4* tp = reinterpret_cast<const float4*>(sc.tris + (cur & ~PT_LEAF_BIT));
                    const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2], r0 = tp[3], r1 = tp[4], r2 = tp[5];
                    const uint32_t top = peek();
                    PT_PRIO_DOWN;
                    float A, B, t;
                    bool bf;
                    const bool ok = ray_triangle(o, d, f3{q0.x, q0.y, q0.z}, f3{q0.w, q1.x, q1.y}, f3{q1.z, q1.w, q2.x},
                                                 (flags & F_CULL) != 0, A, B, t, bf);
                    bool stop = false;
                    if (ok && !(t < 0.0f) && !(t >= h_t)) {
                        h_A = A; h_B = B; h_t = t; h_hit = true; h_back = bf;
                        h_tri = __float_as_uint(q2.y);
                        stop = (flags & F_ANY) != 0;
                    }
                    uint32_t left_after = __float_as_uint(q2.z) - 1u;  // triangles of the leaf after the first
                    if (!stop && left_after != 0u) {
                        if (COUNT) ++cnt.tris;
                        const bool ok2 = ray_triangle(o, d, f3{r0.x, r0.y, r0.z}, f3{r0.w, r1.x, r1.y}, f3{r1.z, r1.w, r2.x},
                                                      (flags & F_CULL) != 0, A, B, t, bf);
                        if (ok2 && !(t < 0.0f) && !(t >= h_t)) {
                            h_A = A; h_B = B; h_t = t; h_hit = true; h_back = bf;
                            h_tri = __float_as_uint(r2.y);
                            stop = (flags & F_ANY) != 0;
                        }
                        left_after -= 1u;
                    }
                    if (stop) {
                        cur = PT_NO_NODE;
                    } else if (left_after != 0u) {
                        cur = cur + 2u;
                    } else {
                        pop_to(top);
                    }
                }
                if (!WIDE && is_leaf(cur)) {
                    if (COUNT) { ++cnt.tris; if (first_active_lane()) ++cnt.wave_tri; }
                    const float4* tp = reinterpret_cast<const float4*>(sc.tris + (cur & ~PT_LEAF_BIT));
                    const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
                    const uint32_t top = peek();
                    float A, B, t;
                    bool bf;
                    const bool ok = ray_triangle(o, d, f3{q0.x, q0.y, q0.z}, f3{q0.w, q1.x, q1.y}, f3{q1.z, q1.w, q2.x},
                                                 (flags & F_CULL) != 0, A, B, t, bf);
                    bool stop = false;
                    if (ok && !(t < 0.0f) && !(t >= h_t)) {
                        h_A = A; h_B = B; h_t = t; h_hit = true; h_back = bf;
                        h_tri = __float_as_uint(q2.y);
                        stop = (flags & F_ANY) != 0;  // any-hit traversal ends at the first accepted triangle
                    }
                    if (stop) {
                        cur = PT_NO_NODE;
                    } else if (__float_as_uint(q2.z) != 1u) {  // triangles of this leaf left, this one included
                        cur = cur + 1u;
                    } else {
                        pop_to(top);
                    }
                }
                if (++burst >= PT_LEAF_BURST) { if (COUNT) tally(3, burst); break; }
                const uint32_t still = __popcll(__ballot(is_leaf(cur)));
                if (still * PT_LEAF_KEEP_DEN < n_leaf * PT_LEAF_KEEP_NUM) { if (COUNT) tally(3, burst); break; }
            }
        } else {
            break;  // no live lane: either the cursor is exhausted or (unreachable) all 64 lanes just refilled nothing
        }
    }
    if (COUNT) {
        atomicAdd(&counters->nodes, static_cast<unsigned long long>(cnt.nodes));
        atomicAdd(&counters->tris, static_cast<unsigned long long>(cnt.tris));
        atomicAdd(&counters->inst_entered, static_cast<unsigned long long>(cnt.inst_entered));
        atomicAdd(&counters->wave_inner, static_cast<unsigned long long>(cnt.wave_inner));
        atomicAdd(&counters->wave_tri, static_cast<unsigned long long>(cnt.wave_tri));
        atomicAdd(&counters->census[1][0], static_cast<unsigned long long>(cnt.inst_tested));
        atomicAdd(&counters->leaf_records, static_cast<unsigned long long>(cnt.leaf_records));
        if (lane == 0)
            for (int ph = 0; ph < 4; ++ph)
                for (int k = 0; k < 6; ++k) atomicAdd(&counters->census[ph][k], static_cast<unsigned long long>(cnt.census[ph][k]));
    }
}
template <bool COUNT, bool WIDE, bool XF_LDS = false, bool LEAF64 = false>
__global__ PT_EXTEND_BOUNDS void traverse_kernel(SceneDev sc, const float4* __restrict__ ray_o,
                                                 const float4* __restrict__ ray_d, uint32_t n, CandQueue cand,
                                                 float4* __restrict__ hit0, uint32_t* __restrict__ hit1,
                                                 uint32_t* __restrict__ spill, uint32_t spill_depth, uint32_t spill_stride,
                                                 uint32_t* __restrict__ next_ray, uint32_t* __restrict__ overflow,
                                                 TraceCounters* counters) {
    traverse_body<COUNT, WIDE, false, XF_LDS, false, LEAF64>(sc, ray_o, ray_d, n, cand, hit0, hit1, spill, spill_depth, spill_stride, next_ray, overflow, counters);
}
// The four-wide kernel at seven wavefronts per SIMD (72 registers, 22 KB of LDS per workgroup).  Scenes that the caches
// hold are limited by the latency of a step, and a seventh wavefront hides more of it (2.5-4 % faster); scenes that are
// limited by the gather path (the 10 M-triangle mesh) and the binary layout run 4-7 % slower that way and keep six.
template <bool COUNT, bool XF_LDS = false, bool XF_MANY = false, bool LEAF64 = false>
__global__ __launch_bounds__(kBlock, PT_SEVEN_WAVES) void traverse_kernel7(SceneDev sc, const float4* __restrict__ ray_o,
                                                              const float4* __restrict__ ray_d, uint32_t n, CandQueue cand,
                                                              float4* __restrict__ hit0, uint32_t* __restrict__ hit1,
                                                              uint32_t* __restrict__ spill, uint32_t spill_depth, uint32_t spill_stride,
                                                              uint32_t* __restrict__ next_ray, uint32_t* __restrict__ overflow,
                                                              TraceCounters* counters) {
    traverse_body<COUNT, true, true, XF_LDS, XF_MANY, LEAF64>(sc, ray_o, ray_d, n, cand, hit0, hit1, spill, spill_depth, spill_stride, next_ray, overflow, counters);
}

// Continuation for rays with more than kCandMax candidate instances (a rarity: that many bounding volumes pierced
// before the nearest hit).  The persistent kernels leave such a ray's record as it stands after its kCandMax-th
// instance and list it; here the remaining instances are found by rescanning the DOPs in the same visiting order and
// traced one ray per lane.  Keeping this out of the persistent kernels keeps a function call out of their loop.
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void extend_overflow_kernel(SceneDev sc, const float4* __restrict__ ray_o,
                                                                 const float4* __restrict__ ray_d, CandQueue cand,
                                                                 const uint32_t* __restrict__ overflow,
                                                                 float4* __restrict__ hit0, uint32_t* __restrict__ hit1,
                                                                 uint32_t* __restrict__ spill, uint32_t spill_depth,
                                                                 uint32_t spill_stride, TraceCounters* counters) {
    __shared__ uint32_t s_stack[kStackLds][kBlock];
    const uint32_t gid = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t count = overflow[0];
    TraceCount cnt{};
    for (uint32_t i = gid; i < count; i += gridDim.x * kBlock) {
        const uint32_t rid = overflow[1u + i];
        const float4 ro = ray_o[rid], rd = ray_d[rid];
        const f3 o = ld3(ro), d = ld3(rd);
        const bool any_hit = ((__float_as_uint(rd.w) >> 8) & 1u) != 0;
        const float4 h0 = hit0[rid];
        const uint32_t h1 = hit1[rid];
        LaneHit best;
        best.hit = (h1 & 0x80000000u) != 0;
        best.t = best.hit ? h0.x : kInfF;
        best.A = h0.y; best.B = h0.z; best.tri = __float_as_uint(h0.w);
        best.object = h1 & 0x3fffffffu;
        best.backface = (h1 & 0x40000000u) != 0;
        uint2 last = cand.e[static_cast<size_t>(kCandMax - 1) * cand.stride + rid];
        last.y &= ~kCandLast;
        for (;;) {
            const uint2 e = tlas_rescan(sc, o, d, any_hit, __uint_as_float(last.x), last.y);
            if (e.y == PT_NO_NODE) break;
            if (!any_hit && !(__uint_as_float(e.x) < best.t)) break;
            if (COUNT) ++cnt.inst_entered;
            if (any_hit) trace_instance<true, COUNT>(sc, e.y, o, d, best, s_stack, spill, spill_depth, spill_stride, gid, cnt);
            else trace_instance<false, COUNT>(sc, e.y, o, d, best, s_stack, spill, spill_depth, spill_stride, gid, cnt);
            if (any_hit && best.hit) break;
            last = e;
        }
        hit0[rid] = make_float4(best.t, best.A, best.B, __uint_as_float(best.tri));
        hit1[rid] = best.hit ? (best.object | (best.backface ? 0x40000000u : 0u) | 0x80000000u) : 0u;
    }
    if (COUNT) {
        atomicAdd(&counters->nodes, static_cast<unsigned long long>(cnt.nodes));
        atomicAdd(&counters->tris, static_cast<unsigned long long>(cnt.tris));
        atomicAdd(&counters->inst_entered, static_cast<unsigned long long>(cnt.inst_entered));
        atomicAdd(&counters->wave_inner, static_cast<unsigned long long>(cnt.wave_inner));
        atomicAdd(&counters->wave_tri, static_cast<unsigned long long>(cnt.wave_tri));
    }
}

// ---------------------------------------------------------------------------------------------
// generate
// ---------------------------------------------------------------------------------------------
// Reserves `wave_total` consecutive queue slots for the calling wavefront with ONE atomic per workgroup: a single
// counter word sustains well under 100 M atomics/s, which one atomic per wavefront would exceed at the
// kernel's ray rate.  Every thread of the workgroup must call it.
template <int BLOCK = kBlock>
__device__ __forceinline__ uint32_t block_reserve(uint32_t wave_total, uint32_t* out_count) {
    __shared__ uint32_t s_tot[BLOCK / 64];
    __shared__ uint32_t s_base;
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0) s_tot[wave] = wave_total;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) t += s_tot[w];
        s_base = t ? atomicAdd(out_count, t) : 0u;
    }
    __syncthreads();
    uint32_t base = s_base;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w)
        if (static_cast<uint32_t>(w) < wave) base += s_tot[w];
    return base;
}

// Reservation through several counters.  One counter word takes ~85 M atomics/s on this part (they are executed one after
// another where the word lives), and a PBR shading launch asks once per 256 rays: 290 K times per 73.6 M-ray launch — which
// is 3.4 ms, the duration of the kernel: whatever else it did, the PBR kernel ran at the pace of its counter (a second atomic
// per workgroup on the same word made it five times slower).  So the workgroups of a launch reserve from kStreams counters
// instead, workgroup b from counter b % kStreams, each counter numbering the rays of its own STREAM 0, 1, 2, ...; the
// streams share the output region chunk-wise — stream k owns the chunks k, k + kStreams, k + 2 kStreams, ... of
// kStreamChunk rays — so a ray's slot follows from its number in its stream by arithmetic alone (stream_slot), with no table
// and no waiting.  What this leaves is a ragged end: the streams do not end in the same chunk and their last chunks are
// not full.  Two small kernels close it after the launch (stream_gather_kernel, stream_scatter_kernel): the valid rays
// beyond the last chunk row that every stream filled are gathered and written back contiguously, and the launch's total
// goes to *out_count.  Everything downstream sees the dense queue it always saw.
#ifndef PT_STREAMS_LOG2
#define PT_STREAMS_LOG2 3
#endif
#ifndef PT_STREAM_CHUNK_LOG2
#define PT_STREAM_CHUNK_LOG2 14
#endif
constexpr uint32_t kStreamsLog2 = PT_STREAMS_LOG2, kStreams = 1u << kStreamsLog2;
constexpr uint32_t kStreamChunkLog2 = PT_STREAM_CHUNK_LOG2, kStreamChunk = 1u << kStreamChunkLog2;
constexpr uint32_t kStreamCursorStride = 32;  // cursors sit 128 bytes apart
__host__ __device__ __forceinline__ uint32_t stream_slot(uint32_t stream, uint32_t number) {
    return ((((number >> kStreamChunkLog2) << kStreamsLog2) | stream) << kStreamChunkLog2) | (number & (kStreamChunk - 1u));
}
struct Reservation {
    uint32_t* out_count;  // the shared counter (streams == nullptr), else written by the fix-up
    uint32_t* streams;    // kStreams cursors, kStreamCursorStride words apart, zeroed before the launch; or null
};
// Reserves `wave_total` consecutive numbers for the calling wavefront; returns the first.  With streams they are numbers in
// the workgroup's stream (turn each into a slot with queue_slot), else slots.  Every thread of the workgroup must call it.
template <int BLOCK = kBlock>
__device__ __forceinline__ uint32_t reserve_slots(uint32_t wave_total, const Reservation& r) {
    return block_reserve<BLOCK>(wave_total, r.streams ? r.streams + (blockIdx.x & (kStreams - 1u)) * kStreamCursorStride : r.out_count);
}
__device__ __forceinline__ uint32_t queue_slot(const Reservation& r, uint32_t number) {
    return r.streams ? stream_slot(blockIdx.x & (kStreams - 1u), number) : number;
}

// The ragged end of a launch reserved through streams.  `row0` = the number of chunk rows that every stream filled
// completely: all slots below row0 * kStreams * kStreamChunk hold rays.  Beyond, a slot holds a ray iff its number in its
// stream is below that stream's cursor.
struct StreamEnd {
    uint32_t first;  // first slot of the ragged zone
    uint32_t end;    // one past its last slot
    uint32_t pos[kStreams];
};
__device__ __forceinline__ StreamEnd stream_end(const uint32_t* __restrict__ streams) {
    StreamEnd e;
    uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
    for (uint32_t k = 0; k < kStreams; ++k) {
        e.pos[k] = streams[k * kStreamCursorStride];
        const uint32_t full = e.pos[k] >> kStreamChunkLog2, used = (e.pos[k] + kStreamChunk - 1u) >> kStreamChunkLog2;
        lo = full < lo ? full : lo;
        hi = used > hi ? used : hi;
    }
    e.first = lo << (kStreamsLog2 + kStreamChunkLog2);
    e.end = hi << (kStreamsLog2 + kStreamChunkLog2);
    return e;
}
// Pass 1: the rays of the zone's slots [first + from, first + from + slots), in any order, to a staging queue (one atomic
// per workgroup on *staged).  If they do not fit, *failed is set and pass 2 leaves the queue as it is.
__global__ __launch_bounds__(kBlock) void stream_gather_kernel(RayQueue q, const uint32_t* __restrict__ streams, RayQueue staging,
                                                               uint32_t staging_capacity, uint32_t* __restrict__ staged,
                                                               uint32_t* __restrict__ failed, uint32_t from, uint32_t slots) {
    const StreamEnd e = stream_end(streams);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t begin = e.first + from;  // (from <= end - first: the launcher's pieces stay inside the zone)
    const uint32_t end = slots < e.end - begin ? begin + slots : e.end;
    for (uint32_t s0 = begin + blockIdx.x * kBlock; s0 < end; s0 += gridDim.x * kBlock) {  // (uniform per workgroup: every thread reaches the barriers)
        const uint32_t s = s0 + threadIdx.x;
        const uint32_t k = (s >> kStreamChunkLog2) & (kStreams - 1u);
        const uint32_t number = ((s >> (kStreamsLog2 + kStreamChunkLog2)) << kStreamChunkLog2) | (s & (kStreamChunk - 1u));
        uint32_t pos = 0;
#pragma unroll
        for (uint32_t j = 0; j < kStreams; ++j) pos = (j == k) ? e.pos[j] : pos;
        const bool valid = s < end && number < pos;
        const unsigned long long m = __ballot(valid);
        const uint32_t base = block_reserve(static_cast<uint32_t>(__popcll(m)), staged);
        if (valid) {
            const uint32_t dst = base + __popcll(m & ((1ull << lane) - 1ull));
            if (dst < staging_capacity) {
                staging.o[dst] = q.o[s];
                staging.d[dst] = q.d[s];
                staging.p[dst] = q.p[s];
            } else {
                atomicOr(failed, 1u);
            }
        }
    }
}
// Pass 2: back, contiguously, from slot first + to on; the total so far to *out_count.
__global__ __launch_bounds__(kBlock) void stream_scatter_kernel(RayQueue q, const uint32_t* __restrict__ streams, RayQueue staging,
                                                                const uint32_t* __restrict__ staged, const uint32_t* __restrict__ failed,
                                                                uint32_t to, uint32_t* __restrict__ out_count) {
    if (*failed) return;  // (uniform) the staged rays are not all of the piece: the launcher goes piece by piece instead
    const StreamEnd e = stream_end(streams);
    const uint32_t n = *staged, first = e.first + to;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        q.o[first + i] = staging.o[i];
        q.d[first + i] = staging.d[i];
        q.p[first + i] = staging.p[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *out_count = first + n;
}

// DENSE: the frame is a whole number of tiles, so every lane has a pixel and ray i goes to slot i — no compaction, no
// queue atomic (one per workgroup is ~100 M/s at this kernel's rate, more than one counter word sustains).
template <bool DENSE>
__global__ __launch_bounds__(kBlock) void generate_kernel(CameraDev cam, const float2* __restrict__ sub,
                                                          uint32_t sample_begin, uint32_t sample_step,
                                                          uint32_t max_bounces, uint32_t seed, uint64_t first, uint32_t n,
                                                          uint32_t slots_in_batch, uint32_t tile_begin, uint32_t tile_step,
                                                          uint32_t n_tiles, RayQueue out, uint32_t* __restrict__ out_count) {
    // n_tiles = tiles rendered here: tile_begin, tile_begin + tile_step, ... (all of them when 0 / 1)
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    bool valid = false;
    float4 o4, d4, p4;
    if (i < n) {
        uint32_t lane, tile, slot;
        if (slots_in_batch) {
            // batch = sample slots [first, first + slots_in_batch) of every tile, TILE-major: all samples of a tile
            // are neighbours in the queue, so a contiguous part of the queue is a compact part of the screen
            const uint32_t per_tile = slots_in_batch * 64u;
            tile = i / per_tile;
            const uint32_t rem = i - tile * per_tile;
            slot = static_cast<uint32_t>(first) + (rem >> 6);
            lane = rem & 63u;
        } else {  // linear sample-major index (slot * n_tiles + tile) * 64 + lane, starting at `first`
            const uint64_t g = first + i;
            lane = static_cast<uint32_t>(g & 63u);
            const uint64_t tile_lin = g >> 6;
            tile = static_cast<uint32_t>(tile_lin % n_tiles);
            slot = static_cast<uint32_t>(tile_lin / n_tiles);
        }
        tile = tile_begin + tile * tile_step;
        const uint32_t sample = sample_begin + slot * sample_step;
        const int x = static_cast<int>((tile % cam.tiles_x) * kTile + (lane & 7u));
        const int y = static_cast<int>((tile / cam.tiles_x) * kTile + (lane >> 3));
        if (x < cam.width && y < cam.height) {
            valid = true;
            const float2 s = sub[sample];
            // ndc_from_pixel + generate_camera_ray (renderer.cc:17-21, 66-75)
            const float px = static_cast<float>(x) + s.x, py = static_cast<float>(y) + s.y;
            const float nx = px * cam.inv_w * 2.0f - 1.0f;
            const float ny = -py * cam.inv_h * 2.0f + 1.0f;
            const float* m = cam.ndc2world;
            f3 dir;  // xyz of mat4 * (nx, ny, 0, 1): (c0*x + c1*y) + (c2*z + c3*w)
            dir.x = (m[0] * nx + m[4] * ny) + (m[8] * 0.0f + m[12] * 1.0f);
            dir.y = (m[1] * nx + m[5] * ny) + (m[9] * 0.0f + m[13] * 1.0f);
            dir.z = (m[2] * nx + m[6] * ny) + (m[10] * 0.0f + m[14] * 1.0f);
            dir = normalize3(dir);
            const uint32_t pid = tile * 64u + lane;
            const uint32_t pixel = static_cast<uint32_t>(y) * cam.width + x;
            o4 = make_float4(cam.origin[0], cam.origin[1], cam.origin[2], __uint_as_float(pid));
            // bits 0-7 bounces left, bit 8 any-hit, bits 9-31 the framebuffer slot again (frames below 2^23 slots): the
            // PBR shade kernel then needs no read of the origin record
            // (larger frames keep those bits clear: the PBR kernel keeps a pass-through ray's tag there, see pass_hash)
            const bool slot_fits = static_cast<uint32_t>(cam.tiles_x) * static_cast<uint32_t>(cam.tiles_y) * 64u < (1u << 23);
            d4 = make_float4(dir.x, dir.y, dir.z, __uint_as_float((max_bounces & 0xffu) | (slot_fits ? pid << 9 : 0u)));
            p4 = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(rng_camera_key(seed, pixel, sample)));
        }
    }
    if (DENSE) {
        if (valid) {
            out.o[i] = o4;
            out.d[i] = d4;
            out.p[i] = p4;
        }
        if (i == 0) *out_count = n;
        return;
    }
    // compaction: rays of a wave stay contiguous and ordered
    const unsigned long long mask = __ballot(valid);
    const uint32_t lane_id = threadIdx.x & 63u;
    const uint32_t base = block_reserve(static_cast<uint32_t>(__popcll(mask)), out_count);
    if (valid) {
        const uint32_t dst = base + __popcll(mask & ((1ull << lane_id) - 1ull));
        out.o[dst] = o4;
        out.d[dst] = d4;
        out.p[dst] = p4;
    }
}

// ---------------------------------------------------------------------------------------------
// textures and materials (cpu_framebuffer.h:35-50, 90-123; brdf.cc:11-69)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int wrap_coord(int v, int n) {  // ((v % n) + n) % n
    if ((n & (n - 1)) == 0) return v & (n - 1);  // power-of-two sizes: the same value without an integer division
    int r = v % n;
    return (r < 0) ? r + n : r;
}
struct Taps {
    int x0, y0, x1, y1;
    float tx, ty;
};
__device__ __forceinline__ Taps bilinear_taps(float u, float v, int w, int h) {
    float xf = u * static_cast<float>(w) - 0.5f, yf = v * static_cast<float>(h) - 0.5f;
    float fx = floorf(xf), fy = floorf(yf);
    Taps t;
    t.tx = xf - fx;
    t.ty = yf - fy;
    int x0 = static_cast<int>(fx), y0 = static_cast<int>(fy);
    t.x0 = wrap_coord(x0, w);
    t.y0 = wrap_coord(y0, h);
    // wrap(x0 + 1) follows from wrap(x0): the next texel of the row, or texel 0 after the last one
    t.x1 = (t.x0 + 1 == w) ? 0 : t.x0 + 1;
    t.y1 = (t.y0 + 1 == h) ? 0 : t.y0 + 1;
    return t;
}
// The sRGB decode table (256 floats) is copied to LDS by every shading workgroup: a texture sample decodes twelve
// channels, and as global gathers those would be half of the kernel's vector memory instructions.
// Its neighbour holds i / 255 (the linear decode and every alpha channel): a table read instead of an IEEE division per
// channel, the same binary32 value.
__shared__ float s_srgb_lut[256];
__shared__ float s_unorm_lut[256];
// The first kObjLds object records (matrices + material, 256 B each) are read by nearly every hit: a copy in LDS takes
// those reads (15 of a PBR hit's ~58 loads) off the vector memory path.
constexpr uint32_t kObjLds = PT_OBJ_LDS;
__shared__ float4 s_objects[kObjLds * (sizeof(DevObject) / 16)];
__device__ __forceinline__ void stage_decode_tables(const SceneDev& sc) {
    static_assert(kBlock >= 256, "one table entry per thread (of the first 256)");
    if (threadIdx.x < 256u) {
        s_srgb_lut[threadIdx.x] = sc.srgb_lut[threadIdx.x];
        s_unorm_lut[threadIdx.x] = static_cast<float>(threadIdx.x) / 255.0f;
    }
    const uint32_t n_quarters = (sc.n_objects < kObjLds ? sc.n_objects : kObjLds) * (sizeof(DevObject) / 16);
    for (uint32_t q = threadIdx.x; q < n_quarters; q += blockDim.x) s_objects[q] = reinterpret_cast<const float4*>(sc.objects)[q];
    __syncthreads();
}
// ALL_LDS: the scene has at most kObjLds objects, so every record is in the LDS copy.  The pointer is then known to be an
// LDS address: its reads wait for the LDS counter only, instead of a flat read's wait for every load in flight.
template <bool ALL_LDS = false>
__device__ __forceinline__ const DevObject* object_record(const SceneDev& sc, uint32_t object) {
    if (ALL_LDS) return reinterpret_cast<const DevObject*>(s_objects) + (object < kObjLds ? object : 0u);
    return (object < kObjLds) ? reinterpret_cast<const DevObject*>(s_objects) + object : sc.objects + object;
}
__device__ __forceinline__ float4 texel8(const SceneDev& sc, uint32_t word, bool srgb) {
    uint32_t r = word & 0xffu, g = (word >> 8) & 0xffu, b = (word >> 16) & 0xffu, a = word >> 24;
    const float fa = s_unorm_lut[a];
    if (srgb) return make_float4(s_srgb_lut[r], s_srgb_lut[g], s_srgb_lut[b], fa);
    return make_float4(s_unorm_lut[r], s_unorm_lut[g], s_unorm_lut[b], fa);
}
__device__ __forceinline__ float4 sample_image(const SceneDev& sc, const DevTexture& im, float u, float v, bool srgb) {
    const Taps t = bilinear_taps(u, v, im.width, im.height);
    const uint32_t* px = sc.texels + im.offset;
    const uint32_t w00 = px[t.y0 * im.width + t.x0], w10 = px[t.y0 * im.width + t.x1], w01 = px[t.y1 * im.width + t.x0],
                   w11 = px[t.y1 * im.width + t.x1];
    const float4 top = mix4(texel8(sc, w00, srgb), texel8(sc, w10, srgb), t.tx);
    const float4 bot = mix4(texel8(sc, w01, srgb), texel8(sc, w11, srgb), t.tx);
    return mix4(top, bot, t.ty);
}
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 sample_albedo(const SceneDev& sc, const DevMaterial& m, float u, float v) {
    float4 f = make_float4(m.base_color[0], m.base_color[1], m.base_color[2], m.base_color[3]);
    if (m.base_color_tex.width) f = mul4(f, sample_image(sc, m.base_color_tex, u, v, true));
    return f;
}
__device__ __forceinline__ float4 sample_orm(const SceneDev& sc, const DevMaterial& m, float u, float v) {
    float4 f = make_float4(1.0f, m.roughness, m.metallic, 1.0f);
    if (m.mr_tex.width) f = mul4(f, sample_image(sc, m.mr_tex, u, v, false));
    return f;
}
__device__ __forceinline__ float4 sample_normals(const SceneDev& sc, const DevMaterial& m, float u, float v) {
    if (m.normal_tex.width) return sample_image(sc, m.normal_tex, u, v, false);
    return make_float4(0.5f, 0.5f, 1.0f, 0.0f);
}
__device__ __forceinline__ f3 sample_emissive(const SceneDev& sc, const DevMaterial& m, float u, float v) {
    float4 f = make_float4(m.emissive[0], m.emissive[1], m.emissive[2], 1.0f);
    if (m.emissive_tex.width) f = mul4(f, sample_image(sc, m.emissive_tex, u, v, true));
    return f3{f.x * m.emissive_strength, f.y * m.emissive_strength, f.z * m.emissive_strength};
}
__device__ __forceinline__ float sample_transmission(const SceneDev& sc, const DevMaterial& m, float u, float v) {
    float f = m.transmission;
    if (m.transmission_tex.width) f *= sample_image(sc, m.transmission_tex, u, v, false).x;
    return f;
}
__device__ __forceinline__ f3 sample_environment(const SceneDev& sc, f3 dir) {  // brdf.cc:62-69
    float u = (pt_atan2(-dir.z, -dir.x) + sc.env_rot) * (1.0f / kPiF);
    float v = (-2.0f * pt_asin(dir.y)) * (1.0f / kPiF);
    u = u * 0.5f + 0.5f;
    v = v * 0.5f + 0.5f;
    const Taps t = bilinear_taps(u, v, sc.env_w, sc.env_h);
    const float4 c00 = sc.env[t.y0 * sc.env_w + t.x0], c10 = sc.env[t.y0 * sc.env_w + t.x1],
                 c01 = sc.env[t.y1 * sc.env_w + t.x0], c11 = sc.env[t.y1 * sc.env_w + t.x1];
    const float4 r = mix4(mix4(c00, c10, t.tx), mix4(c01, c11, t.tx), t.ty);
    return f3{r.x, r.y, r.z};
}

// ---------------------------------------------------------------------------------------------
// shading helpers (ray_program.cc:16-142)
// ---------------------------------------------------------------------------------------------
struct VertexD {
    f3 position, normal;
    float4 tangent;
    float u, v;
};
__device__ __forceinline__ VertexD load_vertex(const float4* __restrict__ p) {
    const float4 a = p[0], b = p[1], c = p[2];
    VertexD r;
    r.position = f3{a.x, a.y, a.z};
    r.normal = f3{a.w, b.x, b.y};
    r.tangent = make_float4(b.z, b.w, c.x, c.y);
    r.u = c.z;
    r.v = c.w;
    return r;
}
__device__ __forceinline__ float lin3(float a, float b, float c, float A, float B, float C) { return a * A + b * B + c * C; }
__device__ __forceinline__ f3 lin3(f3 a, f3 b, f3 c, float A, float B, float C) { return a * A + b * B + c * C; }
__device__ __forceinline__ f3 cos_dir(float xi_x, float xi_y) {  // ImportanceSampleCosDir
    float c = sqrtf(1.0f - xi_x), s = sqrtf(xi_x);
    float sp, cp;
    pt_sincos(2.0f * xi_y * kPiF, &sp, &cp);
    return f3{s * cp, s * sp, c};
}
__device__ __forceinline__ f3 ggx_dir(float xi_x, float xi_y, float a) {  // importanceSampleGGX
    float c2 = (1.0f - xi_x) / (1.0f + (a * a - 1.0f) * xi_x);
    float c = sqrtf(c2), s = sqrtf(1.0f - c2);
    float sp, cp;
    pt_sincos(2.0f * xi_y * kPiF, &sp, &cp);
    return f3{s * cp, s * sp, c};
}
__device__ __forceinline__ f3 fresnel_schlick(f3 f0, f3 f90, float c) {
    float x = 1.0f - c;
    float x2 = x * x;
    float x5 = x2 * x2 * x;
    return mix3(f0, f90, x5);
}
__device__ __forceinline__ float v_smith_ggx(float NoV, float NoL, float a) {  // V_SmithGGXCorrelated (2/(..): sic)
    float a2 = a * a;
    float gv = NoL * sqrtf((-NoV * a2 + NoV) * NoV + a2);
    float gl = NoV * sqrtf((-NoL * a2 + NoL) * NoL + a2);
    return 2.0f / (gv + gl);
}
__device__ __forceinline__ f3 reflect3(f3 I, f3 N) { return I - N * dot3(N, I) * 2.0f; }
__device__ __forceinline__ f3 refract3(f3 I, f3 N, float eta) {
    float d = dot3(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k >= 0.0f) return eta * I - (eta * d + sqrtf(k)) * N;
    return mk3(0.0f);
}
struct Tbn {
    f3 t, b, n;
};
__device__ __forceinline__ Tbn make_tbn(f3 tangent, f3 bitangent, f3 normal) {  // construct_TBN
    Tbn r;
    r.n = normalize3(normal);
    r.t = normalize3(tangent - r.n * dot3(r.n, tangent));
    r.b = normalize3(bitangent - r.n * dot3(r.n, bitangent) - r.t * dot3(r.t, bitangent));
    return r;
}
__device__ __forceinline__ f3 normal_map_to_world(float4 s, const Tbn& tbn) {  // normal_map_sample_to_world
    f3 n = f3{fmaf(s.x, 2.0f, -1.0f), fmaf(s.y, 2.0f, -1.0f), fmaf(s.z, 2.0f, -1.0f)};
    n.y *= -1.0f;
    return mul_tbn(tbn.t, tbn.b, tbn.n, n);
}

struct ShadePoint {
    VertexD p;           // interpolated vertex; normal/tangent in world space after to_world()
    f3 p1, p2, p3;       // object-space corner positions (geometric normal)
    f3 bitangent;
    const DevObject* obj;
};
// get_vertex_data + interpolate_vertex_data + convert_normals_to_world_space (ray_program.cc:75-101)
template <bool ALL_LDS = false>
__device__ __forceinline__ ShadePoint fetch_point(const SceneDev& sc, uint32_t object, uint32_t tri, float A, float B) {
    ShadePoint sp;
    sp.obj = object_record<ALL_LDS>(sc, object);
    const float4* rec = sc.corners + static_cast<size_t>(sp.obj->tri_base + tri / 3u) * 9u;  // tri = 3 * triangle number
    const VertexD v1 = load_vertex(rec), v2 = load_vertex(rec + 3), v3 = load_vertex(rec + 6);
    const float C = 1.0f - A - B;
    sp.p.position = lin3(v1.position, v2.position, v3.position, A, B, C);
    sp.p.normal = lin3(v1.normal, v2.normal, v3.normal, A, B, C);
    sp.p.tangent = make_float4(lin3(v1.tangent.x, v2.tangent.x, v3.tangent.x, A, B, C),
                               lin3(v1.tangent.y, v2.tangent.y, v3.tangent.y, A, B, C),
                               lin3(v1.tangent.z, v2.tangent.z, v3.tangent.z, A, B, C),
                               lin3(v1.tangent.w, v2.tangent.w, v3.tangent.w, A, B, C));
    sp.p.u = lin3(v1.u, v2.u, v3.u, A, B, C);
    sp.p.v = lin3(v1.v, v2.v, v3.v, A, B, C);
    sp.p1 = v1.position; sp.p2 = v2.position; sp.p3 = v3.position;
    return sp;
}
__device__ __forceinline__ void normals_to_world(ShadePoint& sp) {
    const float sign = sp.p.tangent.w;
    const f3 t = xform(sp.obj->model, f3{sp.p.tangent.x, sp.p.tangent.y, sp.p.tangent.z}, 0.0f);
    sp.p.tangent = make_float4(t.x, t.y, t.z, 0.0f);
    sp.p.normal = xform(sp.obj->normal, sp.p.normal, 0.0f);
    sp.bitangent = cross3(sp.p.normal, t) * sign;
}

// append of up to NSLOT rays per lane: slot-major order inside a wave keeps rays of one lobe contiguous
template <int NSLOT>
__device__ __forceinline__ void append_rays(const bool (&has)[NSLOT], const float4 (&o)[NSLOT], const float4 (&d)[NSLOT],
                                            const float4 (&p)[NSLOT], RayQueue out, uint32_t* out_count) {
    const uint32_t lane_id = threadIdx.x & 63u;
    const unsigned long long lt = (1ull << lane_id) - 1ull;
    uint32_t offs[NSLOT];
    uint32_t total = 0;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const unsigned long long m = __ballot(has[s]);
        offs[s] = total + __popcll(m & lt);
        total += __popcll(m);
    }
    const uint32_t base = block_reserve(total, out_count);
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        if (has[s]) {
            const uint32_t dst = base + offs[s];
            out.o[dst] = o[s];
            out.d[dst] = d[s];
            out.p[dst] = p[s];
        }
    }
}

__device__ __forceinline__ void add_radiance(FrameDev fb, uint32_t pid, f3 c) {
    if (c.x != 0.0f || c.y != 0.0f || c.z != 0.0f) {
        // the bucket: binades 2^-40 .. 2^8 of the largest component in eight steps of six (smaller and larger ones, and
        // contributions that are not numbers, go to the ends)
        if constexpr (kFrameBuckets == 8) {
        // (magnitudes compare like their bit patterns)
        const uint32_t mx = __float_as_uint(c.x) & 0x7fffffffu, my = __float_as_uint(c.y) & 0x7fffffffu, mz = __float_as_uint(c.z) & 0x7fffffffu;
        const uint32_t m = mx > my ? (mx > mz ? mx : mz) : (my > mz ? my : mz);
        int e = static_cast<int>(m >> 23) - (127 - 40);
        e = e < 0 ? 0 : (e > 47 ? 47 : e);
        pid += ((static_cast<uint32_t>(e) * 43u) >> 8) * 3u * fb.stride;  // e / 6 for 0 <= e < 48; an index below 2^32
        }
        atomicAdd(fb.sum + pid, c.x);
        atomicAdd(fb.sum + pid + fb.stride, c.y);
        atomicAdd(fb.sum + pid + 2u * fb.stride, c.z);
    }
}

// ---------------------------------------------------------------------------------------------
// shade: RayCaster (ray_program.cc:150-163)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void shade_raycast_kernel(SceneDev sc, RayQueue rays, uint32_t n, HitQueue hits,
                                                               FrameDev fb) {
    stage_decode_tables(sc);
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 ro = rays.o[i], rd = rays.d[i];
    const uint32_t h1 = hits.h1[i];
    f3 c;
    if (!(h1 & 0x80000000u)) {
        c = sample_environment(sc, ld3(rd));
    } else {
        const float4 h0 = hits.h0[i];  // written for hits only
        const uint32_t object = h1 & 0x3fffffffu;
        const DevObject* obj = object_record(sc, object);
        const float4* rec = sc.corners + static_cast<size_t>(obj->tri_base + __float_as_uint(h0.w) / 3u) * 9u;
        const float4 c1 = rec[2], c2 = rec[5], c3 = rec[8];
        const float A = h0.y, B = h0.z, C = 1.0f - A - B;
        const float u = lin3(c1.z, c2.z, c3.z, A, B, C), v = lin3(c1.w, c2.w, c3.w, A, B, C);
        const float4 a = sample_albedo(sc, obj->mat, u, v);
        c = f3{a.x, a.y, a.z};
    }
    add_radiance(fb, __float_as_uint(ro.w), c);
}

// ---------------------------------------------------------------------------------------------
// shade: AmbientOcclusion (ray_program.cc:172-225)
// ---------------------------------------------------------------------------------------------
template <bool ALL_LDS>  // every object record of the scene is in the LDS copy (see object_record)
__global__ __launch_bounds__(kBlock) void shade_ao_kernel(SceneDev sc, RayQueue rays, uint32_t n, HitQueue hits,
                                                          RayQueue out, uint32_t* __restrict__ out_count, FrameDev fb,
                                                          uint32_t ao_samples) {
    stage_decode_tables(sc);
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    bool spawn = false;
    f3 new_pos = mk3(0.0f);
    Tbn tbn{mk3(0.0f), mk3(0.0f), mk3(0.0f)};
    float jx = 0.0f, jy = 0.0f;
    uint32_t pid = 0, key = 0, new_depth = 0;
    if (i < n) {
        const float4 ro = rays.o[i], rd = rays.d[i], rp = rays.p[i];
        const uint32_t h1 = hits.h1[i];
        pid = __float_as_uint(ro.w);
        key = __float_as_uint(rp.w);
        const uint32_t depth = __float_as_uint(rd.w) & 0xffu;
        if (!(h1 & 0x80000000u)) {
            add_radiance(fb, pid, f3{rp.x, rp.y, rp.z});  // miss: return the payload
        } else if (depth != 0) {
            const float4 h0 = hits.h0[i];  // written for hits only
            ShadePoint sp = fetch_point<ALL_LDS>(sc, h1 & 0x3fffffffu, __float_as_uint(h0.w), h0.y, h0.z);
            normals_to_world(sp);
            const DevMaterial& mat = sp.obj->mat;
            if ((h1 & 0x40000000u) && mat.double_sided) sp.p.normal = sp.p.normal * -1.0f;
            tbn = make_tbn(f3{sp.p.tangent.x, sp.p.tangent.y, sp.p.tangent.z}, sp.bitangent, sp.p.normal);
            const float4 nm = sample_normals(sc, mat, sp.p.u, sp.p.v);
            if (nm.w != 0.0f) {
                const f3 nv = normal_map_to_world(nm, tbn);
                tbn = make_tbn(tbn.t, tbn.b, nv);
            }
            const f3 ws_pos = ld3(ro) + ld3(rd) * h0.x;
            new_pos = ws_pos + sp.p.normal * 1e-5f;
            new_depth = (depth - 1u) & 0xffu;
            jx = rng_uniform(key, 0);
            jy = rng_uniform(key, 1);
            spawn = true;
        }
    }
    // every spawning lane emits ao_samples rays: wave-level reservation, sample-major order
    const unsigned long long mask = __ballot(spawn);
    const uint32_t lane_id = threadIdx.x & 63u;
    const uint32_t n_spawn = __popcll(mask);
    const uint32_t base = block_reserve(n_spawn * ao_samples, out_count);
    if (!spawn) return;
    const uint32_t rank = __popcll(mask & ((1ull << lane_id) - 1ull));
    const float inv = 1.0f / static_cast<float>(ao_samples);
    for (uint32_t s = 0; s < ao_samples; ++s) {
        // fibonacci2D + Cranley-Patterson shift (ray_program.cc:212-215; brdf.cc:73-78)
        const float fi = static_cast<float>(s);
        float rx = (fi + 0.5f) * inv;
        float gy = (fi + 1.0f) * 1.618034f;
        float ry = gy - floorf(gy);  // fmod(x, 1) for x >= 0
        rx = rx + jx; rx = rx - floorf(rx);
        ry = ry + jy; ry = ry - floorf(ry);
        const f3 dir = mul_tbn(tbn.t, tbn.b, tbn.n, cos_dir(rx, ry));
        const uint32_t dst = base + s * n_spawn + rank;
        out.o[dst] = make_float4(new_pos.x, new_pos.y, new_pos.z, __uint_as_float(pid));
        out.d[dst] = make_float4(dir.x, dir.y, dir.z, __uint_as_float(new_depth | 0x100u | (pid << 9)));
        out.p[dst] = make_float4(inv, inv, inv, __uint_as_float(rng_child(key, LOBE_AO0 + s)));
    }
}

// ---------------------------------------------------------------------------------------------
// shade: PBR (ray_program.cc:234-391)
// ---------------------------------------------------------------------------------------------
// Threads per workgroup of the PBR shading kernel.  One queue-reservation atomic per workgroup: at 256 threads a launch of
// 74 M rays issues 290 K of them in 3.7 ms, close to what one counter word sustains (~85 M/s) — but larger workgroups are
// slower all the same (512: +5 %, 1024: +24 %: the two barriers of the reservation weigh more), so 256 it stays.
#ifndef PT_SHADE_BLOCK
#define PT_SHADE_BLOCK 256
#endif
constexpr int kShadeBlock = PT_SHADE_BLOCK;
#ifdef PT_SHADE_WAVES
#define PT_SHADE_BOUNDS __launch_bounds__(kShadeBlock, PT_SHADE_WAVES)
#else
#define PT_SHADE_BOUNDS __launch_bounds__(kShadeBlock)
#endif
// Largest component of a payload as a selection weight; NaN, infinite and zero payloads are never chosen.
__device__ __forceinline__ float lobe_weight(f3 p) {
    float w = fabsf(p.x);
    if (fabsf(p.y) > w) w = fabsf(p.y);
    if (fabsf(p.z) > w) w = fabsf(p.z);
    return ((w > 0.0f) && (w < kInfF)) ? w : 0.0f;
}

// A pass-through ray (alpha != 1, ray_program.cc:265-271) starts at position -/+ normal * 1e-5 with the INTERPOLATED normal.
// Where that normal and the triangle's plane disagree about the side the ray came from (flipped vertex normals; smooth normals
// under strong non-uniform scale at a low angle; coordinates so large that the offset vanishes), the new origin is in front
// of the triangle again, the ray hits it again a hair's breadth away, and the reference's loop (renderer.cc:111-117) never
// ends — with the other lobes spawned at every step.  Here a pass-through ray carries, in the nine bits above the framebuffer
// slot of its origin record (frames below 2^23 slots) or above the depth of its direction record (larger frames): a 5-bit
// tag of the triangle it passed, the number (1..7) of pass-through steps in a row that ended closer than kPassNear, and one
// bit "this ray was already a repeated visit"; and, as the sign of its payload's first component (never negative otherwise),
// a mark "that number is not 0".  Only a marked ray whose hit is that close again reads the nine bits, so ordinary frames
// never pay for them.  A chain ends — no pass-through ray, one more in FrameDev::abandoned — at the third visit in a row of one
// triangle (the common case, cut early because every step spawns its lobes), and in any case at the eighth close step in a
// row (a ray on an edge may alternate between the triangles that share it).  Eight real layers less than kPassNear apart
// leave nothing of a payload worth following; different triangles that close with equal tags twice running: 1e-3.
constexpr float kPassNear = 1e-3f;
constexpr uint32_t kPassSteps = 7u;
__device__ __forceinline__ uint32_t pass_hash(uint32_t object, uint32_t triangle) {
    return ((triangle * 0x9e3779b9u) ^ (object * 0x85ebca6bu)) >> 27;
}

template <bool SINGLE, bool ALL_LDS>
__global__ PT_SHADE_BOUNDS void shade_pbr_kernel(SceneDev sc, RayQueue rays, uint32_t n, HitQueue hits,
                                                           RayQueue out, Reservation res, FrameDev fb) {
    // Three stages, so that no spawned ray has to wait in registers for the others:
    //   A  everything up to the point where it is known WHICH of the four lobes spawn a ray (that needs the tangent
    //      frame, the micro-normal and the refracted direction);
    //   B  one queue reservation for the workgroup (ballots + one atomic);
    //   C  each lobe's direction and weight, written straight to its reserved slot.
    // SINGLE (no counterpart in the reference): the lobes are computed first, one of them is kept — chosen with
    // probability proportional to the largest component of its payload, payload divided by that probability — and
    // the reservation is for that one ray.
    stage_decode_tables(sc);
    const uint32_t i = blockIdx.x * kShadeBlock + threadIdx.x;
    bool has[4] = {false, false, false, false};  // pass-through, diffuse, transmission, reflection
    // state that survives from stage A to stage C
    const float4 ro = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 rd = ro;
    // (the view vector and the refracted direction are cheap to form again in stage C; the diffuse direction is formed
    // in stage A, so that the tangent and bitangent need not survive)
    f3 payload = mk3(0.0f), position = mk3(0.0f), normal = mk3(0.0f), atten = mk3(1.0f), m = mk3(0.0f), l_diff = mk3(0.0f);
    f3 diffuse_color = mk3(0.0f), f0 = mk3(0.0f), shading_n = mk3(0.0f);
    float alpha = 0.0f, transmission = 0.0f, a_lin = 0.0f, eta = 1.0f;
    uint32_t key = 0;
    bool exiting = false;

    // the framebuffer slot rides in the direction record's flag word when it fits (uniform choice per frame); asked for
    // again in stage C instead of being kept in a register
    const bool slot_fits = fb.stride < (1u << 23);
    auto slot_of = [&]() -> uint32_t {
        return slot_fits ? (__float_as_uint(rd.w) >> 9) : __float_as_uint(rays.o[i].w);
    };
    uint32_t pass_tag = 0;  // of the pass-through ray this hit spawns (see pass_hash)
    if (i < n) {
        rd = rays.d[i];
        const float4 rp = rays.p[i];
        const uint32_t h1 = hits.h1[i];
        const uint32_t pid = slot_of();
        key = __float_as_uint(rp.w);
        const uint32_t depth = __float_as_uint(rd.w) & 0xffu;
        // (a payload is never negative while base colours, alpha and F0 stay within glTF's [0, 1]: the sign of its first component
        // marks a pass-through ray spawned at a close hit, see kPassNear)
        const bool marked = (__float_as_uint(rp.x) >> 31) != 0u;
        payload = f3{fabsf(rp.x), rp.y, rp.z};
        const f3 dir = ld3(rd);
        // (radiance is added where it is known, not after the branches have joined: three registers less across the lobes' set-up)
        if (!(h1 & 0x80000000u)) {
            add_radiance(fb, pid, payload * sample_environment(sc, dir));
        } else {
            const bool backface = (h1 & 0x40000000u) != 0;
            const float4 h0 = hits.h0[i];  // written for hits only
            ShadePoint sp = fetch_point<ALL_LDS>(sc, h1 & 0x3fffffffu, __float_as_uint(h0.w), h0.y, h0.z);
            sp.p.position = xform(sp.obj->model, sp.p.position, 1.0f);
            normals_to_world(sp);
            const DevMaterial& mat = sp.obj->mat;
            if (backface) {
                if (mat.double_sided) sp.p.normal = sp.p.normal * -1.0f;
                else if (mat.has_volume) exiting = true;
            }
            position = sp.p.position;
            normal = sp.p.normal;
            const float4 albedo = sample_albedo(sc, mat, sp.p.u, sp.p.v);
            alpha = albedo.w;
            if (!mat.alpha_blending) alpha = (alpha < mat.alpha_cutoff) ? 0.0f : 1.0f;
            if (alpha != 1.0f) {  // pass-through ray: same direction and depth (ray_program.cc:265-271)
                const uint32_t h = pass_hash(h1 & 0x3fffffffu, __float_as_uint(h0.w));
                const bool near = h0.x < kPassNear;
                uint32_t tag = 0;  // (read once in a blue moon)
                if (near && marked) tag = slot_fits ? (__float_as_uint(rays.o[i].w) >> 23) : ((__float_as_uint(rd.w) >> 9) & 0x1ffu);
                const uint32_t close_steps = (tag >> 5) & 7u;  // (0 for a ray that is not marked)
                const bool again = near && marked && ((tag & 0x1fu) == h);
                if (near && ((again && (tag >> 8)) || close_steps >= kPassSteps)) atomicAdd(fb.abandoned, 1u);  // the chain does not advance
                else pass_tag = (again ? 0x100u : 0u) | ((near ? close_steps + 1u : 0u) << 5) | h, has[0] = true;
            }
            if (alpha != 0.0f) {
                if (exiting) {  // volume absorption on the segment that leaves the volume (:278-279)
                    const f3 e = f3{-mat.attenuation[0], -mat.attenuation[1], -mat.attenuation[2]} * h0.x;
                    atten = f3{sel_min(pt_exp(e.x), 1.0f), sel_min(pt_exp(e.y), 1.0f), sel_min(pt_exp(e.z), 1.0f)};
                }
                const f3 emissive = sample_emissive(sc, mat, sp.p.u, sp.p.v);
                add_radiance(fb, pid, payload * emissive * atten * alpha);
                if (depth != 0) {
                    const f3 v = -dir;
                    const float4 nm = sample_normals(sc, mat, sp.p.u, sp.p.v);
                    // handle_TBN_creation (ray_program.cc:121-142)
                    Tbn tbn = make_tbn(f3{sp.p.tangent.x, sp.p.tangent.y, sp.p.tangent.z}, sp.bitangent, sp.p.normal);
                    const bool has_map = (nm.w != 0.0f);
                    f3 nv = has_map ? normal_map_to_world(nm, tbn) : tbn.n;
                    const bool impossible = dot3(exiting ? -v : v, nv) < 0.0f;
                    if (impossible) {
                        f3 g = normalize3(cross3(sp.p2 - sp.p1, sp.p3 - sp.p1));
                        if (mat.double_sided && backface) g = g * -1.0f;
                        nv = xform(sp.obj->normal, g, 0.0f);
                    }
                    if (has_map || impossible) tbn = make_tbn(tbn.t, tbn.b, nv);

                    transmission = sample_transmission(sc, mat, sp.p.u, sp.p.v);
                    const float4 orm = sample_orm(sc, mat, sp.p.u, sp.p.v);
                    const f3 alb = f3{albedo.x, albedo.y, albedo.z};
                    diffuse_color = (1.0f - orm.z) * alb;
                    f0 = mix3(mk3(mat.f0), alb, orm.z);
                    a_lin = orm.y * orm.y;
                    const float u0 = rng_uniform(key, 0), u1 = rng_uniform(key, 1);
                    const float u2 = rng_uniform(key, 2), u3 = rng_uniform(key, 3);
                    // micro-normal shared by transmission and reflection (:324-389)
                    m = mul_tbn(tbn.t, tbn.b, tbn.n, ggx_dir(u2, u3, a_lin));
                    if (exiting) m = m * -1.0f;
                    eta = (!exiting) ? 1.0f / mat.ior : mat.ior;
                    const f3 l_refr = refract3(-v, m, eta);
                    shading_n = tbn.n;
                    has[1] = any_ne(diffuse_color * (1.0f - transmission), mk3(0.0f));
                    if (has[1]) l_diff = normalize3(mul_tbn(tbn.t, tbn.b, tbn.n, cos_dir(u0, u1)));
                    has[2] = any_ne(diffuse_color * transmission, mk3(0.0f)) && any_ne(l_refr, mk3(0.0f));
                    has[3] = true;
                }
            }
        }
    }

    const uint32_t lane_id = threadIdx.x & 63u;
    const unsigned long long lt = (1ull << lane_id) - 1ull;
    auto store = [&](uint32_t number, f3 org, float4 d4, f3 pl, uint32_t lobe) {
        const uint32_t dst = queue_slot(res, number);
        const bool pass = lobe == LOBE_PASS;
        out.o[dst] = make_float4(org.x, org.y, org.z, __uint_as_float(slot_of() | ((pass && slot_fits) ? (pass_tag & 0x1ffu) << 23 : 0u)));
        if (pass && !slot_fits) d4.w = __uint_as_float((__float_as_uint(d4.w) & 0x1ffu) | ((pass_tag & 0x1ffu) << 9));
        out.d[dst] = d4;
        const float plx = (pass && (pass_tag & 0xe0u)) ? -pl.x : pl.x;  // the mark: spawned at a close hit (see stage A)
        out.p[dst] = make_float4(plx, pl.y, pl.z, __uint_as_float(rng_child(key, lobe)));
    };
    // ---- stage C: the lobes; `emit(s, ...)` is called with s = the lobe's number (LOBE_PASS .. LOBE_REFLECT) ----
    auto lobes = [&](auto&& emit) {
    if (has[0]) emit(std::integral_constant<int, 0>{}, position + (exiting ? 1.0f : -1.0f) * normal * 1e-5f, rd, payload * (1.0f - alpha), LOBE_PASS);
    if (has[3]) {  // every lane that reaches the lobes has the reflection lobe
        const f3 f90 = mk3(1.0f);
        const f3 v = -ld3(rd);
        const f3 l_refr = refract3(-v, m, eta);
        const uint32_t nflags = (((__float_as_uint(rd.w) & 0xffu) - 1u) & 0xffu) | (slot_fits ? slot_of() << 9 : 0u);
        if (has[1]) {  // diffuse lobe (:306-323)
            const f3 l = l_diff;
            const f3 h = normalize3(v + l);
            const float LdH = clampf(dot3(l, h), 0.0f, 1.0f);
            const f3 F = mk3(1.0f) - fresnel_schlick(f0, f90, LdH);
            const f3 pl = F * diffuse_color * (1.0f - transmission) * atten * payload * alpha;
            emit(std::integral_constant<int, 1>{}, position + normal * 1e-5f, make_float4(l.x, l.y, l.z, __uint_as_float(nflags)), pl, LOBE_DIFFUSE);
        }
        f3 N = shading_n;
        f3 offset_dir = normal;
        if (exiting) {
            N = N * -1.0f;
            offset_dir = offset_dir * -1.0f;
        }
        const float VdN = clampf(dot3(N, v), 1e-8f, 1.0f);
        const float VdM = clampf(dot3(v, m), 0.0f, 1.0f);
        const float LdM = clampf(dot3(l_refr, -m), 0.0f, 1.0f);
        const f3 F = fresnel_schlick(f0, f90, (!exiting) ? VdM : LdM);
        if (has[2]) {  // transmission (:345-370)
            const f3 h = normalize3(-(eta * v + l_refr));
            const float LdN = clampf(dot3(-N, l_refr), 0.0f, 1.0f);
            const float G = v_smith_ggx(VdN, LdN, a_lin);
            const float NdM = clampf(dot3(N, m), 1e-8f, 1.0f);
            const float VdH = clampf(dot3((!exiting) ? v : -v, h), 0.0f, 1.0f);
            const float w = G * VdH * LdN / NdM;
            f3 brdf = (mk3(1.0f) - F) * w;
            brdf = f3{sel_min(brdf.x, 10.0f), sel_min(brdf.y, 10.0f), sel_min(brdf.z, 10.0f)};
            const f3 ln = normalize3(l_refr);
            const f3 pl = diffuse_color * transmission * atten * brdf * payload * alpha;
            emit(std::integral_constant<int, 2>{}, position - offset_dir * 1e-5f, make_float4(ln.x, ln.y, ln.z, __uint_as_float(nflags)), pl, LOBE_TRANSMIT);
        }
        {   // specular reflection (:371-388)
            const f3 h = m;
            const f3 l = normalize3(reflect3(-v, h));
            const float LdH = clampf(dot3(l, h), 0.0f, 1.0f);
            const float LdN = clampf(dot3(N, l), 0.0f, 1.0f);
            const float NdH = clampf(dot3(N, h), 1e-8f, 1.0f);
            const float G = v_smith_ggx(VdN, LdN, a_lin);
            const float w = G * LdN * LdH / NdH;
            f3 brdf = F * w;
            brdf = f3{sel_min(brdf.x, 10.0f), sel_min(brdf.y, 10.0f), sel_min(brdf.z, 10.0f)};
            const f3 pl = atten * brdf * payload * alpha;
            emit(std::integral_constant<int, 3>{}, position + offset_dir * 1e-5f, make_float4(l.x, l.y, l.z, __uint_as_float(nflags)), pl, LOBE_REFLECT);
        }
    }
    };

    if constexpr (!SINGLE) {
        // ---- stage B: slots for the rays of this workgroup; lobe-major inside a wavefront ----
        uint32_t offs[4];
        uint32_t total = 0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const unsigned long long mask = __ballot(has[s]);
            offs[s] = total + __popcll(mask & lt);
            total += __popcll(mask);
        }
        const uint32_t base = reserve_slots<kShadeBlock>(total, res);
        lobes([&](auto s, f3 org, float4 d4, f3 pl, uint32_t lobe) { store(base + offs[decltype(s)::value], org, d4, pl, lobe); });
    } else {
        f3 org0 = mk3(0.0f), org1 = org0, org2 = org0, org3 = org0, pl0 = org0, pl1 = org0, pl2 = org0, pl3 = org0;
        float4 d0 = ro, d1 = ro, d2 = ro, d3 = ro;
        lobes([&](auto s, f3 org, float4 d4, f3 pl, uint32_t) {
            constexpr int k = decltype(s)::value;
            if constexpr (k == 0) { org0 = org; d0 = d4; pl0 = pl; }
            if constexpr (k == 1) { org1 = org; d1 = d4; pl1 = pl; }
            if constexpr (k == 2) { org2 = org; d2 = d4; pl2 = pl; }
            if constexpr (k == 3) { org3 = org; d3 = d4; pl3 = pl; }
        });
        // cumulative weights in lobe order; an absent lobe has weight 0 and is never the first to exceed x
        const float w0 = has[0] ? lobe_weight(pl0) : 0.0f, w1 = has[1] ? lobe_weight(pl1) : 0.0f;
        const float w2 = has[2] ? lobe_weight(pl2) : 0.0f, w3 = has[3] ? lobe_weight(pl3) : 0.0f;
        const float c0 = w0, c1 = c0 + w1, c2 = c1 + w2, total_w = c2 + w3;
        const bool any = total_w > 0.0f;
        int pick = 3;
        if (any) {
            const float x = rng_uniform(key, 4) * total_w;
            if (x < c2) pick = 2;
            if (x < c1) pick = 1;
            if (x < c0) pick = 0;
            // only when x landed on the total by rounding: the last lobe with a weight
            if (pick == 3 && w3 == 0.0f) pick = (w2 != 0.0f) ? 2 : ((w1 != 0.0f) ? 1 : 0);
        }
        const float wp = (pick == 0) ? w0 : (pick == 1) ? w1 : (pick == 2) ? w2 : w3;
        const f3 org = (pick == 0) ? org0 : (pick == 1) ? org1 : (pick == 2) ? org2 : org3;
        const float4 d4 = (pick == 0) ? d0 : (pick == 1) ? d1 : (pick == 2) ? d2 : d3;
        const f3 pl = (pick == 0) ? pl0 : (pick == 1) ? pl1 : (pick == 2) ? pl2 : pl3;
        const unsigned long long mask = __ballot(any);
        const uint32_t base = reserve_slots<kShadeBlock>(__popcll(mask), res);
        if (any) store(base + __popcll(mask & lt), org, d4, pl * (total_w / wp), static_cast<uint32_t>(pick));
    }
}

// ---------------------------------------------------------------------------------------------
// framebuffer
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void clear_kernel(float4* __restrict__ p, uint64_t n4) {
    uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (; i < n4; i += stride) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ __launch_bounds__(kBlock) void resolve_kernel(FrameDev fb, CameraDev cam, float4* __restrict__ out, float divisor,
                                                         float alpha) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t npix = static_cast<uint32_t>(cam.width) * cam.height;
    if (i >= npix) return;
    const uint32_t x = i % cam.width, y = i / cam.width;
    const uint32_t tile = (y / kTile) * cam.tiles_x + (x / kTile);
    const uint32_t pid = tile * 64u + (y % kTile) * 8u + (x % kTile);
    float r = 0.0f, g = 0.0f, b = 0.0f;
    for (uint32_t k = 0; k < kFrameBuckets; ++k) {  // smallest magnitudes first
        const float* sum = fb.sum + static_cast<size_t>(k) * 3u * fb.stride;
        r += sum[pid];
        g += sum[fb.stride + pid];
        b += sum[2u * fb.stride + pid];
    }
    out[i] = make_float4(r / divisor, g / divisor, b / divisor, alpha);
}
__global__ __launch_bounds__(kBlock) void divide_rgba_kernel(float4* __restrict__ p, uint64_t n, float divisor) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    float4 v = p[i];
    p[i] = make_float4(v.x / divisor, v.y / divisor, v.z / divisor, 1.0f);
}
__global__ __launch_bounds__(kBlock) void pack_rays_kernel(const float* __restrict__ o, const float* __restrict__ d,
                                                           const uint8_t* __restrict__ any, uint32_t n, RayQueue out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    out.o[i] = make_float4(o[3 * i], o[3 * i + 1], o[3 * i + 2], __uint_as_float(i));
    out.d[i] = make_float4(d[3 * i], d[3 * i + 1], d[3 * i + 2], __uint_as_float((any && any[i]) ? 0x100u : 0u));
    out.p[i] = make_float4(1.f, 1.f, 1.f, 0.f);
}

// hit queue records -> the ABI's 32-byte hit records (t, A, B, object, mesh, triangle, hit, backface), on the device
__global__ __launch_bounds__(kBlock) void unpack_hits_kernel(const float4* __restrict__ h0, const uint32_t* __restrict__ h1,
                                                             const uint32_t* __restrict__ object_mesh, uint32_t n,
                                                             uint4* __restrict__ out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 a = h0[i];
    const uint32_t b = h1[i];
    const bool hit = (b & 0x80000000u) != 0;
    const uint32_t object = hit ? (b & 0x3fffffffu) : 0u;
    out[2 * i] = make_uint4(__float_as_uint(hit ? a.x : kInfF), __float_as_uint(hit ? a.y : 0.0f), __float_as_uint(hit ? a.z : 0.0f), object);
    out[2 * i + 1] = make_uint4(hit ? object_mesh[object] : 0u, hit ? __float_as_uint(a.w) : 0u, hit ? 1u : 0u,
                                (hit && (b & 0x40000000u)) ? 1u : 0u);
}

inline uint32_t blocks_for(uint64_t n) { return static_cast<uint32_t>((n + kBlock - 1) / kBlock); }

// A launch that the runtime rejects (too much LDS, too many registers, a bad grid) returns nothing: the error is only in
// hipGetLastError, and a later synchronise succeeds as if the kernel had run.  Every launcher below therefore asks right
// after the launch and keeps the first failure, with the kernel's name, for the caller to collect.
thread_local hipError_t g_launch_error = hipSuccess;
thread_local const char* g_launch_kernel = "";
inline void note_launch(const char* kernel) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess && g_launch_error == hipSuccess) {
        g_launch_error = e;
        g_launch_kernel = kernel;
    }
}
#define PT_LAUNCH(kernel, ...)                      \
    do {                                            \
        hipLaunchKernelGGL(kernel, __VA_ARGS__);    \
        note_launch(#kernel);                       \
    } while (0)

}  // namespace

hipError_t take_launch_error(const char** kernel) {
    const hipError_t e = g_launch_error;
    if (kernel) *kernel = g_launch_kernel;
    g_launch_error = hipSuccess;
    g_launch_kernel = "";
    return e;
}

bool generate_is_dense(const CameraDev& cam) { return cam.width % kTile == 0 && cam.height % kTile == 0; }
void launch_generate(hipStream_t st, const CameraDev& cam, const float2* subsamples, uint32_t sample_begin,
                     uint32_t sample_step, uint32_t max_bounces, uint32_t seed, uint64_t first, uint32_t n,
                     uint32_t slots_in_batch, TileSubset tiles, RayQueue out, uint32_t* out_count) {
    if (!n) return;
    if (generate_is_dense(cam))
        PT_LAUNCH(generate_kernel<true>, dim3(blocks_for(n)), dim3(kBlock), 0, st, cam, subsamples, sample_begin,
                           sample_step, max_bounces, seed, first, n, slots_in_batch, tiles.begin, tiles.step, tiles.count, out, out_count);
    else
        PT_LAUNCH(generate_kernel<false>, dim3(blocks_for(n)), dim3(kBlock), 0, st, cam, subsamples, sample_begin,
                           sample_step, max_bounces, seed, first, n, slots_in_batch, tiles.begin, tiles.step, tiles.count, out, out_count);
}
void launch_extend(hipStream_t st, const SceneDev& sc, RayQueue rays, uint32_t n, HitQueue hits, uint32_t* spill,
                   uint32_t spill_depth, uint32_t spill_stride, TraceCounters* counters) {
    if (!n) return;
    if (counters)
        PT_LAUNCH(extend_kernel<true>, dim3(blocks_for(n)), dim3(kBlock), 0, st, sc, rays.o, rays.d, n, hits.h0,
                           hits.h1, spill, spill_depth, spill_stride, counters);
    else
        PT_LAUNCH(extend_kernel<false>, dim3(blocks_for(n)), dim3(kBlock), 0, st, sc, rays.o, rays.d, n, hits.h0,
                           hits.h1, spill, spill_depth, spill_stride, counters);
}
uint32_t extend_persistent_grid(int device, bool seven_waves) {
    int per_cu = 0, cus = 0;
    const hipError_t e = seven_waves ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, traverse_kernel7<false, false>, kBlock, 0)
                                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, traverse_kernel<false, false>, kBlock, 0);
    if (e != hipSuccess || per_cu < 1)
        per_cu = 2;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
    return static_cast<uint32_t>(per_cu) * static_cast<uint32_t>(cus);
}
void launch_extend_persistent(hipStream_t st, const SceneDev& sc, RayQueue rays, uint32_t n, HitQueue hits, CandQueue cand,
                              uint32_t* spill, uint32_t spill_depth, uint32_t n_blocks, uint32_t* next_ray, uint32_t* overflow,
                              TraceCounters* counters, hipEvent_t between) {
    if (!n) return;
    const uint32_t grid = std::min<uint32_t>(n_blocks, blocks_for(n));
    const uint32_t stride = n_blocks * kBlock;  // one spill column per persistent thread
    for (hipError_t e : {hipMemsetAsync(next_ray, 0, 8 * kCursorStride * sizeof(uint32_t), st), hipMemsetAsync(overflow, 0, sizeof(uint32_t), st)})
        if (e != hipSuccess && g_launch_error == hipSuccess) {
            g_launch_error = e;
            g_launch_kernel = "hipMemsetAsync of the traversal cursors";
        }
    const dim3 g_all(blocks_for(n)), g_p(grid), g_c(std::min<uint32_t>(n_blocks, 64u)), b(kBlock);
    cand.h1 = hits.h1;
    if (counters) {
        if (sc.tlas) PT_LAUNCH(tlas_tree_kernel<true>, g_all, b, 0, st, sc, rays.o, rays.d, n, cand, counters);
        else PT_LAUNCH(tlas_kernel<true>, g_all, b, 0, st, sc, rays.o, rays.d, n, cand, counters);
        if (between) (void)hipEventRecord(between, st);
        // (the counting form always runs as the six-workgroup kernel: its counters do not fit the seven-wave register budget)
        if (sc.nodes4 && sc.leaves64)
            PT_LAUNCH((traverse_kernel<true, true, false, true>), g_p, b, 0, st, sc, rays.o, rays.d, n, cand, hits.h0, hits.h1, spill,
                               spill_depth, stride, next_ray, overflow, counters);
        else if (sc.nodes4)
            PT_LAUNCH((traverse_kernel<true, true>), g_p, b, 0, st, sc, rays.o, rays.d, n, cand, hits.h0, hits.h1, spill,
                               spill_depth, stride, next_ray, overflow, counters);
        else
            PT_LAUNCH((traverse_kernel<true, false>), g_p, b, 0, st, sc, rays.o, rays.d, n, cand, hits.h0, hits.h1, spill,
                               spill_depth, stride, next_ray, overflow, counters);
        PT_LAUNCH(extend_overflow_kernel<true>, g_c, b, 0, st, sc, rays.o, rays.d, cand, overflow, hits.h0, hits.h1, spill,
                           spill_depth, stride, counters);
    } else {
        if (sc.tlas) PT_LAUNCH(tlas_tree_kernel<false>, g_all, b, 0, st, sc, rays.o, rays.d, n, cand, counters);
        else PT_LAUNCH(tlas_kernel<false>, g_all, b, 0, st, sc, rays.o, rays.d, n, cand, counters);
        if (between) (void)hipEventRecord(between, st);
        // (the forms with the last argument set read the instance transforms from LDS only: scenes whose transforms all fit there)
        if (PT_XF_LDS && sc.nodes4 && sc.leaves64 && sc.seven_waves && sc.n_instances <= kXformLds7)
            PT_LAUNCH((traverse_kernel7<false, true, false, true>), g_p, b, 0, st, sc, rays.o, rays.d, n, cand, hits.h0, hits.h1, spill, spill_depth,
                               stride, next_ray, overflow, counters);
        else if (PT_XF_LDS && sc.nodes4 && sc.leaves64 && sc.seven_waves && sc.n_instances <= 2u * kXformLds7)
            PT_LAUNCH((traverse_kernel7<false, true, true, true>), g_p, b, 0, st, sc, rays.o, rays.d, n, cand, hits.h0, hits.h1, spill, spill_depth,
                               stride, next_ray, overflow, counters);
        else if (sc.nodes4 && sc.leaves64 && sc.seven_waves)
            PT_LAUNCH((traverse_kernel7<false, false, false, true>), g_p, b, 0, st, sc, rays.o, rays.d, n, cand, hits.h0, hits.h1, spill, spill_depth,
                               stride, next_ray, overflow, counters);
        else if (PT_XF_LDS && sc.nodes4 && sc.leaves64 && sc.n_instances <= kXformLds)
            PT_LAUNCH((traverse_kernel<false, true, true, true>), g_p, b, 0, st, sc, rays.o, rays.d, n, cand, hits.h0, hits.h1, spill,
                               spill_depth, stride, next_ray, overflow, counters);
        else if (sc.nodes4 && sc.leaves64)
            PT_LAUNCH((traverse_kernel<false, true, false, true>), g_p, b, 0, st, sc, rays.o, rays.d, n, cand, hits.h0, hits.h1, spill,
                               spill_depth, stride, next_ray, overflow, counters);
        else if (PT_XF_LDS && sc.nodes4 && sc.seven_waves && sc.n_instances <= kXformLds7)
            PT_LAUNCH((traverse_kernel7<false, true>), g_p, b, 0, st, sc, rays.o, rays.d, n, cand, hits.h0, hits.h1, spill, spill_depth,
                               stride, next_ray, overflow, counters);
        else if (PT_XF_LDS && sc.nodes4 && sc.seven_waves && sc.n_instances <= 2u * kXformLds7)
            PT_LAUNCH((traverse_kernel7<false, true, true>), g_p, b, 0, st, sc, rays.o, rays.d, n, cand, hits.h0, hits.h1, spill, spill_depth,
                               stride, next_ray, overflow, counters);
        else if (sc.nodes4 && sc.seven_waves)
            PT_LAUNCH((traverse_kernel7<false, false>), g_p, b, 0, st, sc, rays.o, rays.d, n, cand, hits.h0, hits.h1, spill, spill_depth,
                               stride, next_ray, overflow, counters);
        else if (PT_XF_LDS && sc.nodes4 && sc.n_instances <= kXformLds)
            PT_LAUNCH((traverse_kernel<false, true, true>), g_p, b, 0, st, sc, rays.o, rays.d, n, cand, hits.h0, hits.h1, spill,
                               spill_depth, stride, next_ray, overflow, counters);
        else if (sc.nodes4)
            PT_LAUNCH((traverse_kernel<false, true, false>), g_p, b, 0, st, sc, rays.o, rays.d, n, cand, hits.h0, hits.h1, spill,
                               spill_depth, stride, next_ray, overflow, counters);
        else if (PT_XF_LDS && sc.n_instances <= kXformLds)
            PT_LAUNCH((traverse_kernel<false, false, true>), g_p, b, 0, st, sc, rays.o, rays.d, n, cand, hits.h0, hits.h1, spill,
                               spill_depth, stride, next_ray, overflow, counters);
        else
            PT_LAUNCH((traverse_kernel<false, false>), g_p, b, 0, st, sc, rays.o, rays.d, n, cand, hits.h0, hits.h1, spill,
                               spill_depth, stride, next_ray, overflow, counters);
        // continuation of the rare rays with more candidate instances than the lists hold (returns at once when none)
        PT_LAUNCH(extend_overflow_kernel<false>, g_c, b, 0, st, sc, rays.o, rays.d, cand, overflow, hits.h0, hits.h1, spill,
                           spill_depth, stride, counters);
    }
}
uint32_t shade_stream_slack() { return kStreams * (kStreamChunk + 8u * kShadeBlock); }
void launch_shade(hipStream_t st, int program, const SceneDev& sc, RayQueue rays, uint32_t n, HitQueue hits, RayQueue out,
                  uint32_t* out_count, FrameDev fb, uint32_t ao_samples, bool single_path, const ShadeStreams* streams) {
    if (!n) return;
    const dim3 grid(blocks_for(n)), block(kBlock);
    const bool all_lds = sc.n_objects <= kObjLds;  // every object record is in the kernel's LDS copy
    if (program == 1) {
        if (all_lds) PT_LAUNCH(shade_ao_kernel<true>, grid, block, 0, st, sc, rays, n, hits, out, out_count, fb, ao_samples);
        else PT_LAUNCH(shade_ao_kernel<false>, grid, block, 0, st, sc, rays, n, hits, out, out_count, fb, ao_samples);
    } else if (program == 2) {
        const dim3 g((n + kShadeBlock - 1) / kShadeBlock), b(kShadeBlock);
        // launches of few workgroups are far from the counter's rate: they keep the single counter and need no fix-up
        const bool use_streams = streams && streams->cursors && g.x >= 4096u;
        const Reservation res{out_count, use_streams ? streams->cursors : nullptr};
        if (use_streams) {
            const hipError_t e = hipMemsetAsync(streams->cursors, 0, kShadeStreamWords * sizeof(uint32_t), st);
            if (e != hipSuccess && g_launch_error == hipSuccess) {
                g_launch_error = e;
                g_launch_kernel = "hipMemsetAsync of the reservation cursors";
            }
        }
        if (single_path && all_lds) PT_LAUNCH((shade_pbr_kernel<true, true>), g, b, 0, st, sc, rays, n, hits, out, res, fb);
        else if (single_path) PT_LAUNCH((shade_pbr_kernel<true, false>), g, b, 0, st, sc, rays, n, hits, out, res, fb);
        else if (all_lds) PT_LAUNCH((shade_pbr_kernel<false, true>), g, b, 0, st, sc, rays, n, hits, out, res, fb);
        else PT_LAUNCH((shade_pbr_kernel<false, false>), g, b, 0, st, sc, rays, n, hits, out, res, fb);
        if (use_streams) {  // close the ragged end: the last two of the cursor words are the staged count and the failure flag
            uint32_t* staged = streams->cursors + kShadeStreamWords - 2u;
            const dim3 gf(1024);
            PT_LAUNCH(stream_gather_kernel, gf, block, 0, st, out, streams->cursors, streams->staging, streams->staging_capacity, staged, staged + 1,
                      0u, 0xffffffffu);
            PT_LAUNCH(stream_scatter_kernel, gf, block, 0, st, out, streams->cursors, streams->staging, staged, staged + 1, 0u, out_count);
        }
    } else
        PT_LAUNCH(shade_raycast_kernel, grid, block, 0, st, sc, rays, n, hits, fb);
}
bool shade_reserves_through_streams(int program, uint32_t n, const ShadeStreams* streams) {
    return program == 2 && streams && streams->cursors && (n + kShadeBlock - 1) / kShadeBlock >= 4096u;
}
hipError_t close_ragged_end_piecewise(hipStream_t st, RayQueue out, uint32_t* out_count, const ShadeStreams& streams, uint32_t* total) {
    // The streams ended so far apart that the ragged zone's rays did not fit the staging queue in one piece (pass 2 has
    // left the queue alone).  The zone is compacted in place, in pieces of as many SLOTS as the staging queue holds rays:
    // a piece's rays are staged, then written where the previous piece's ended — never past the piece's own first slot.
    uint32_t pos[kStreams];
    for (uint32_t k = 0; k < kStreams; ++k) {
        const hipError_t e = hipMemcpyAsync(&pos[k], streams.cursors + k * kStreamCursorStride, 4, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) return e;
    }
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    uint32_t lo = 0xffffffffu, hi = 0u;
    for (uint32_t k = 0; k < kStreams; ++k) {
        lo = std::min(lo, pos[k] >> kStreamChunkLog2);
        hi = std::max(hi, (pos[k] + kStreamChunk - 1u) >> kStreamChunkLog2);
    }
    const uint32_t zone = (hi - lo) << (kStreamsLog2 + kStreamChunkLog2);
    uint32_t* staged = streams.cursors + kShadeStreamWords - 2u;
    const dim3 gf(1024), block(kBlock);
    uint32_t written = 0;  // rays written back so far = offset of the next piece's destination in the zone
    for (uint32_t from = 0; from < zone; from += streams.staging_capacity) {
        if ((e = hipMemsetAsync(staged, 0, 8, st)) != hipSuccess) return e;  // the staged count and the failure flag
        PT_LAUNCH(stream_gather_kernel, gf, block, 0, st, out, streams.cursors, streams.staging, streams.staging_capacity, staged, staged + 1, from,
                  streams.staging_capacity);
        PT_LAUNCH(stream_scatter_kernel, gf, block, 0, st, out, streams.cursors, streams.staging, staged, staged + 1, written, out_count);
        uint32_t n = 0;
        if ((e = hipMemcpyAsync(&n, staged, 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
        written += n;
    }
    *total = (lo << (kStreamsLog2 + kStreamChunkLog2)) + written;  // (what the last piece's pass 2 wrote to *out_count)
    return hipSuccess;
}
void launch_clear(hipStream_t st, float* p, uint64_t n) {
    if (!n) return;
    const uint64_t n4 = n / 4;  // callers pad to a multiple of 4 floats
    uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>((n4 + kBlock - 1) / kBlock, 2048));
    PT_LAUNCH(clear_kernel, dim3(grid ? grid : 1), dim3(kBlock), 0, st, reinterpret_cast<float4*>(p), n4);
}
void launch_resolve(hipStream_t st, FrameDev fb, const CameraDev& cam, float* out_rgba, float divisor, float alpha) {
    const uint64_t npix = static_cast<uint64_t>(cam.width) * cam.height;
    PT_LAUNCH(resolve_kernel, dim3(blocks_for(npix)), dim3(kBlock), 0, st, fb, cam,
                       reinterpret_cast<float4*>(out_rgba), divisor, alpha);
}
void launch_divide_rgba(hipStream_t st, float* rgba, uint64_t n_pixels, float divisor) {
    if (!n_pixels) return;
    PT_LAUNCH(divide_rgba_kernel, dim3(blocks_for(n_pixels)), dim3(kBlock), 0, st, reinterpret_cast<float4*>(rgba),
                       n_pixels, divisor);
}
void launch_pack_rays(hipStream_t st, const float* origins, const float* directions, const uint8_t* any_hit, uint32_t n,
                      RayQueue out) {
    if (!n) return;
    PT_LAUNCH(pack_rays_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, st, origins, directions, any_hit, n, out);
}

void launch_unpack_hits(hipStream_t st, HitQueue hits, const uint32_t* object_mesh, uint32_t n, void* out_records) {
    if (!n) return;
    PT_LAUNCH(unpack_hits_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, st, hits.h0, hits.h1, object_mesh, n,
              static_cast<uint4*>(out_records));
}

}  // namespace ptd
