// pt_lbvh.hip — device-side construction of a mesh BVH (linear BVH over Morton-ordered triangles).
//
// No counterpart in the reference, whose only builder is the recursive binned-SAH build on the host
// (BVH_AS::MeshBVHData::parse, acceleration_structure.cc:406-536; restated in pt_bvh.cc and still the default).
// This one trades tree quality for build time on multi-million-triangle meshes:
//   1. 63-bit Morton key of every triangle's centroid (21 bits per axis inside the centroid bounds);
//   2. radix sort of (key, triangle) pairs;
//   3. the binary radix tree over the sorted keys, one thread per inner node (Karras 2012: each node finds its
//      key range and split from common-prefix lengths alone; equal keys are told apart by their position);
//   4. boxes bottom-up: every triangle walks towards the root, the second thread to reach a node merges the
//      children's boxes (exact min / max of vertex coordinates, so the boxes are as tight as the host builder's);
//   5. subtrees of at most `max_leaf` triangles collapse into leaves (a subtree is a contiguous run of the sorted
//      order), the remaining inner nodes are numbered by a prefix sum and written in the builder's node format.
// The result is read back into the same MeshTree the host builder fills, so everything after it (relocation into the
// scene arrays, statistics, conversion to the traversal layout) is shared.  The closest hit along a ray does not
// depend on the tree, so hit records equal the host-built tree's except where two triangles tie in distance.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cfloat>
#include <cstring>
#include <string>

#include "pt_bvh.h"

namespace pth {
namespace {

constexpr uint32_t kNone = 0xffffffffu;
constexpr int kThreads = 256;

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <typename T>
    T* as() const {
        return static_cast<T*>(p);
    }
};

struct Box6 {
    float lo[3], hi[3];
};

__device__ __forceinline__ uint64_t spread21(uint32_t v) {  // bit k of v -> bit 3k
    uint64_t x = v & 0x1fffffu;
    x = (x | (x << 32)) & 0x001f00000000ffffull;
    x = (x | (x << 16)) & 0x001f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

__global__ __launch_bounds__(kThreads) void morton_kernel(const ptd::DevTri* __restrict__ tris, uint32_t n, float3 lo, float3 inv_extent,
                                                         uint64_t* __restrict__ keys, uint32_t* __restrict__ order) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const ptd::DevTri t = tris[i];
    auto cell = [](float s, float lo_, float inv) {
        const float q = (s - lo_) * inv * 2097152.0f;  // 2^21 cells
        return static_cast<uint32_t>(fminf(fmaxf(q, 0.0f), 2097151.0f));
    };
    const uint32_t x = cell(t.p1[0] + t.p2[0] + t.p3[0], lo.x, inv_extent.x);
    const uint32_t y = cell(t.p1[1] + t.p2[1] + t.p3[1], lo.y, inv_extent.y);
    const uint32_t z = cell(t.p1[2] + t.p2[2] + t.p3[2], lo.z, inv_extent.z);
    keys[i] = (spread21(x) << 2) | (spread21(y) << 1) | spread21(z);
    order[i] = i;
}

// length of the common prefix of the keys at sorted positions i and j; equal keys continue with the positions
__device__ __forceinline__ int prefix(const uint64_t* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t a = keys[i], b = keys[j];
    return (a == b) ? 64 + __clz(static_cast<uint32_t>(i) ^ static_cast<uint32_t>(j)) : __clzll(static_cast<long long>(a ^ b));
}

// One thread per inner node of the binary radix tree over n sorted keys (n - 1 inner nodes, node 0 = root).
// child references: bit 31 set = sorted position of a single triangle, clear = inner node.
__global__ __launch_bounds__(kThreads) void radix_tree_kernel(const uint64_t* __restrict__ keys, int n, uint32_t* __restrict__ left,
                                                             uint32_t* __restrict__ right, uint32_t* __restrict__ first,
                                                             uint32_t* __restrict__ last, uint32_t* __restrict__ parent,
                                                             uint32_t* __restrict__ leaf_parent) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (prefix(keys, n, i, i + 1) - prefix(keys, n, i, i - 1)) < 0 ? -1 : 1;
    const int d_min = prefix(keys, n, i, i - d);
    int l_max = 2;
    while (prefix(keys, n, i, i + l_max * d) > d_min) l_max *= 2;
    int l = 0;
    for (int t = l_max / 2; t >= 1; t /= 2)
        if (prefix(keys, n, i, i + (l + t) * d) > d_min) l += t;
    const int j = i + l * d;
    const int d_node = prefix(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (prefix(keys, n, i, i + (s + t) * d) > d_node) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = d > 0 ? i : j, hi = d > 0 ? j : i;
    first[i] = static_cast<uint32_t>(lo);
    last[i] = static_cast<uint32_t>(hi);
    if (lo == gamma) {
        left[i] = PT_LEAF_BIT | static_cast<uint32_t>(gamma);
        leaf_parent[gamma] = static_cast<uint32_t>(i);
    } else {
        left[i] = static_cast<uint32_t>(gamma);
        parent[gamma] = static_cast<uint32_t>(i);
    }
    if (hi == gamma + 1) {
        right[i] = PT_LEAF_BIT | static_cast<uint32_t>(gamma + 1);
        leaf_parent[gamma + 1] = static_cast<uint32_t>(i);
    } else {
        right[i] = static_cast<uint32_t>(gamma + 1);
        parent[gamma + 1] = static_cast<uint32_t>(i);
    }
    if (i == 0) parent[0] = kNone;
}

__device__ __forceinline__ Box6 tri_box(const ptd::DevTri& t) {
    Box6 b;
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = fminf(t.p1[a], fminf(t.p2[a], t.p3[a]));
        b.hi[a] = fmaxf(t.p1[a], fmaxf(t.p2[a], t.p3[a]));
    }
    return b;
}
// Boxes written by other workgroups are read past the (non-coherent) vector L1.
__device__ __forceinline__ Box6 load_box(const Box6* p) {
    Box6 b;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(p);
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = __uint_as_float(__hip_atomic_load(w + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        b.hi[a] = __uint_as_float(__hip_atomic_load(w + 3 + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    return b;
}

// sorted triangles (leaf_count cleared) in Morton order
__global__ __launch_bounds__(kThreads) void gather_kernel(const ptd::DevTri* __restrict__ tris, const uint32_t* __restrict__ order,
                                                         uint32_t n, ptd::DevTri* __restrict__ sorted) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    ptd::DevTri t = tris[order[i]];
    t.leaf_count = 0;
    sorted[i] = t;
}

// One thread per triangle; the second thread to arrive at an inner node owns it and carries on upwards.
__global__ __launch_bounds__(kThreads) void fit_kernel(const ptd::DevTri* __restrict__ sorted, uint32_t n, const uint32_t* __restrict__ left,
                                                      const uint32_t* __restrict__ right, const uint32_t* __restrict__ parent,
                                                      const uint32_t* __restrict__ leaf_parent, uint32_t* __restrict__ arrivals,
                                                      Box6* __restrict__ boxes) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    uint32_t cur = leaf_parent[i];
    while (cur != kNone) {
        __threadfence();  // the box stored below (previous round) is visible before the arrival is counted
        if (atomicAdd(&arrivals[cur], 1u) == 0u) return;
        auto child_box = [&](uint32_t ref) { return (ref & PT_LEAF_BIT) ? tri_box(sorted[ref & ~PT_LEAF_BIT]) : load_box(boxes + ref); };
        const Box6 a = child_box(left[cur]), b = child_box(right[cur]);
        Box6 u;
        for (int k = 0; k < 3; ++k) {
            u.lo[k] = fminf(a.lo[k], b.lo[k]);
            u.hi[k] = fmaxf(a.hi[k], b.hi[k]);
        }
        uint32_t* w = reinterpret_cast<uint32_t*>(boxes + cur);
        for (int k = 0; k < 3; ++k) {
            __hip_atomic_store(w + k, __float_as_uint(u.lo[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(w + 3 + k, __float_as_uint(u.hi[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        cur = parent[cur];
    }
}

__global__ __launch_bounds__(kThreads) void keep_kernel(const uint32_t* __restrict__ first, const uint32_t* __restrict__ last, uint32_t n_inner,
                                                       uint32_t max_leaf, uint32_t* __restrict__ keep) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n_inner) keep[i] = (last[i] - first[i] + 1u > max_leaf) ? 1u : 0u;
}

// One thread per inner node of the radix tree; the kept ones write their record at their compact index.
__global__ __launch_bounds__(kThreads) void emit_kernel(uint32_t n_inner, uint32_t max_leaf, const uint32_t* __restrict__ left,
                                                       const uint32_t* __restrict__ right, const uint32_t* __restrict__ first,
                                                       const uint32_t* __restrict__ last, const uint32_t* __restrict__ keep,
                                                       const uint32_t* __restrict__ compact, const Box6* __restrict__ boxes,
                                                       ptd::DevTri* __restrict__ sorted, ptd::DevNode* __restrict__ nodes) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n_inner || !keep[i]) return;
    ptd::DevNode nd;
    nd.pad0 = nd.pad1 = 0u;
    auto child = [&](uint32_t ref, float* lo, float* hi) -> uint32_t {
        Box6 b;
        uint32_t out;
        if (ref & PT_LEAF_BIT) {  // a single triangle
            const uint32_t pos = ref & ~PT_LEAF_BIT;
            b = tri_box(sorted[pos]);
            sorted[pos].leaf_count = 1u;
            out = ref;
        } else {
            b = boxes[ref];
            if (keep[ref]) {
                out = compact[ref];
            } else {  // a subtree small enough to be one leaf: its run of the sorted order
                out = PT_LEAF_BIT | first[ref];
                sorted[first[ref]].leaf_count = last[ref] - first[ref] + 1u;
            }
        }
        for (int k = 0; k < 3; ++k) {
            lo[k] = b.lo[k];
            hi[k] = b.hi[k];
        }
        return out;
    };
    nd.left = child(left[i], nd.l_lo, nd.l_hi);
    nd.right = child(right[i], nd.r_lo, nd.r_hi);
    nodes[compact[i]] = nd;
}

#define LB_HIP(call)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            if (err) *err = std::string(#call) + " failed: " + hipGetErrorString(e_);           \
            return false;                                                                       \
        }                                                                                       \
    } while (0)
// a kernel launch and the check that the runtime accepted it (a rejected launch is reported by hipGetLastError only)
#define LB_LAUNCH(kernel, ...)                                                                  \
    do {                                                                                        \
        hipLaunchKernelGGL(kernel, __VA_ARGS__);                                                \
        hipError_t e_ = hipGetLastError();                                                      \
        if (e_ != hipSuccess) {                                                                 \
            if (err) *err = std::string("launch of " #kernel " failed: ") + hipGetErrorString(e_); \
            return false;                                                                       \
        }                                                                                       \
    } while (0)

}  // namespace

bool build_mesh_tree_lbvh(const ptgpu_vertex* vertices, const uint32_t* indices, uint32_t n_indices, uint32_t max_leaf, hipStream_t st,
                          MeshTree& out, std::string* err) {
    const uint32_t n = n_indices / 3;
    if (n <= std::max<uint32_t>(max_leaf, 1u)) {  // nothing to split: the host builder's single leaf
        build_mesh_tree(vertices, indices, n_indices, max_leaf, true, false, out);
        return true;
    }
    if (n >= (1u << 29)) {
        if (err) *err = "mesh too large for the device builder";
        return false;
    }
    out.nodes.clear();
    out.tris.assign(n, ptd::DevTri{});
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (uint32_t i = 0; i < n; ++i) {
        ptd::DevTri& t = out.tris[i];
        std::memcpy(t.p1, vertices[indices[3 * i]].position, 12);
        std::memcpy(t.p2, vertices[indices[3 * i + 1]].position, 12);
        std::memcpy(t.p3, vertices[indices[3 * i + 2]].position, 12);
        t.index = 3 * i;
        for (int a = 0; a < 3; ++a) {
            const float s = t.p1[a] + t.p2[a] + t.p3[a];
            if (s < lo[a]) lo[a] = s;
            if (s > hi[a]) hi[a] = s;
        }
    }
    float inv[3];
    for (int a = 0; a < 3; ++a) {
        const float e = hi[a] - lo[a];
        inv[a] = (e > 0.0f && e < FLT_MAX) ? 1.0f / e : 0.0f;
        if (!(lo[a] > -FLT_MAX && lo[a] < FLT_MAX)) lo[a] = 0.0f;  // non-finite coordinates: all cells 0 on this axis
    }

    const uint32_t n_inner = n - 1;
    const uint32_t g_n = (n + kThreads - 1) / kThreads, g_i = (n_inner + kThreads - 1) / kThreads;
    DevBuf d_tris, d_sorted, d_keys, d_keys2, d_order, d_order2, d_left, d_right, d_first, d_last, d_parent, d_leaf_parent, d_arrivals,
        d_boxes, d_keep, d_compact, d_nodes, d_tmp;
    LB_HIP(d_tris.alloc(sizeof(ptd::DevTri) * n));
    LB_HIP(d_sorted.alloc(sizeof(ptd::DevTri) * n));
    LB_HIP(d_keys.alloc(8ull * n));
    LB_HIP(d_keys2.alloc(8ull * n));
    LB_HIP(d_order.alloc(4ull * n));
    LB_HIP(d_order2.alloc(4ull * n));
    for (DevBuf* b : {&d_left, &d_right, &d_first, &d_last, &d_parent, &d_arrivals, &d_keep, &d_compact}) LB_HIP(b->alloc(4ull * n_inner));
    LB_HIP(d_leaf_parent.alloc(4ull * n));
    LB_HIP(d_boxes.alloc(sizeof(Box6) * n_inner));
    LB_HIP(hipMemcpyAsync(d_tris.p, out.tris.data(), sizeof(ptd::DevTri) * n, hipMemcpyHostToDevice, st));

    LB_LAUNCH(morton_kernel, dim3(g_n), dim3(kThreads), 0, st, d_tris.as<ptd::DevTri>(), n, make_float3(lo[0], lo[1], lo[2]),
                       make_float3(inv[0], inv[1], inv[2]), d_keys.as<uint64_t>(), d_order.as<uint32_t>());
    size_t tmp_bytes = 0, scan_bytes = 0;
    LB_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_keys.as<uint64_t>(), d_keys2.as<uint64_t>(), d_order.as<uint32_t>(),
                                               d_order2.as<uint32_t>(), static_cast<int>(n), 0, 63, st));
    LB_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_keep.as<uint32_t>(), d_compact.as<uint32_t>(), static_cast<int>(n_inner), st));
    LB_HIP(d_tmp.alloc(std::max(tmp_bytes, scan_bytes)));
    LB_HIP(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tmp_bytes, d_keys.as<uint64_t>(), d_keys2.as<uint64_t>(), d_order.as<uint32_t>(),
                                               d_order2.as<uint32_t>(), static_cast<int>(n), 0, 63, st));
    LB_LAUNCH(gather_kernel, dim3(g_n), dim3(kThreads), 0, st, d_tris.as<ptd::DevTri>(), d_order2.as<uint32_t>(), n,
                       d_sorted.as<ptd::DevTri>());
    LB_LAUNCH(radix_tree_kernel, dim3(g_i), dim3(kThreads), 0, st, d_keys2.as<uint64_t>(), static_cast<int>(n), d_left.as<uint32_t>(),
                       d_right.as<uint32_t>(), d_first.as<uint32_t>(), d_last.as<uint32_t>(), d_parent.as<uint32_t>(),
                       d_leaf_parent.as<uint32_t>());
    LB_HIP(hipMemsetAsync(d_arrivals.p, 0, 4ull * n_inner, st));
    LB_LAUNCH(fit_kernel, dim3(g_n), dim3(kThreads), 0, st, d_sorted.as<ptd::DevTri>(), n, d_left.as<uint32_t>(), d_right.as<uint32_t>(),
                       d_parent.as<uint32_t>(), d_leaf_parent.as<uint32_t>(), d_arrivals.as<uint32_t>(), d_boxes.as<Box6>());
    LB_LAUNCH(keep_kernel, dim3(g_i), dim3(kThreads), 0, st, d_first.as<uint32_t>(), d_last.as<uint32_t>(), n_inner, max_leaf,
                       d_keep.as<uint32_t>());
    LB_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, scan_bytes, d_keep.as<uint32_t>(), d_compact.as<uint32_t>(), static_cast<int>(n_inner), st));
    uint32_t tail[2] = {0, 0};  // compact index and keep flag of the last radix node -> number of kept nodes
    LB_HIP(hipMemcpyAsync(&tail[0], d_compact.as<uint32_t>() + (n_inner - 1), 4, hipMemcpyDeviceToHost, st));
    LB_HIP(hipMemcpyAsync(&tail[1], d_keep.as<uint32_t>() + (n_inner - 1), 4, hipMemcpyDeviceToHost, st));
    LB_HIP(hipStreamSynchronize(st));
    const uint32_t n_kept = tail[0] + tail[1];
    if (n_kept == 0 || n_kept > n_inner) {
        if (err) *err = "device builder: inconsistent node count";
        return false;
    }
    LB_HIP(d_nodes.alloc(sizeof(ptd::DevNode) * n_kept));
    LB_LAUNCH(emit_kernel, dim3(g_i), dim3(kThreads), 0, st, n_inner, max_leaf, d_left.as<uint32_t>(), d_right.as<uint32_t>(),
                       d_first.as<uint32_t>(), d_last.as<uint32_t>(), d_keep.as<uint32_t>(), d_compact.as<uint32_t>(), d_boxes.as<Box6>(),
                       d_sorted.as<ptd::DevTri>(), d_nodes.as<ptd::DevNode>());
    out.nodes.resize(n_kept);
    LB_HIP(hipMemcpyAsync(out.nodes.data(), d_nodes.p, sizeof(ptd::DevNode) * n_kept, hipMemcpyDeviceToHost, st));
    LB_HIP(hipMemcpyAsync(out.tris.data(), d_sorted.p, sizeof(ptd::DevTri) * n, hipMemcpyDeviceToHost, st));
    LB_HIP(hipStreamSynchronize(st));
    LB_HIP(hipGetLastError());
    out.root = 0u;  // radix node 0 covers all triangles (more than max_leaf), so it is kept and numbered 0
    collect_info(out, n);
    return true;
}

}  // namespace pth
