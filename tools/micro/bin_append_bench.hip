// Microbenchmark: what it costs to append 48-byte ray records (three float4 arrays) to per-cell queues from a shading-like
// kernel, against the plain contiguous append the renderer does (one reservation atomic per 256-thread workgroup).
//   mode 0: contiguous append, one atomic per workgroup
//   mode 1: per-cell append: LDS histogram of the workgroup's cells, one global atomic per (workgroup, non-empty cell),
//           each ray stored at its cell's cursor (cells own fixed regions here: the cheapest possible bookkeeping)
// Cells per ray: random (what the hits of a deep bounce look like) or, with a fourth argument, derived from the workgroup
// index (rays whose origins are still coherent).
// usage: bin_append_bench <mode> <n_cells> <n_rays> [coherent]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void append(float4* __restrict__ o, float4* __restrict__ d, float4* __restrict__ p, uint32_t n,
                                              uint32_t n_cells, uint32_t cell_capacity, uint32_t* __restrict__ cursors, int coherent) {
    extern __shared__ uint32_t s_count[];  // MODE 1: [n_cells] counts, then [n_cells] bases
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < n;
    const uint32_t h = mix(i * 2654435761u + 12345u);
    const uint32_t cell = coherent ? (mix(blockIdx.x >> 4) + (h & 3u)) % n_cells : h % n_cells;
    const float4 vo = make_float4(h, 1.0f, 2.0f, 3.0f), vd = make_float4(0.1f, 0.2f, 0.3f, h), vp = make_float4(1.f, 1.f, 1.f, h);
    if (MODE == 0) {
        __shared__ uint32_t s_base;
        const unsigned long long m = __ballot(valid);
        __shared__ uint32_t s_tot[4];
        if ((threadIdx.x & 63u) == 0) s_tot[threadIdx.x >> 6] = __popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) s_base = atomicAdd(cursors, s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3]);
        __syncthreads();
        uint32_t base = s_base;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) base += s_tot[w];
        if (valid) {
            const uint32_t dst = base + __popcll(m & ((1ull << (threadIdx.x & 63u)) - 1ull));
            o[dst] = vo; d[dst] = vd; p[dst] = vp;
        }
    } else {
        uint32_t* s_base = s_count + n_cells;
        for (uint32_t c = threadIdx.x; c < n_cells; c += 256) s_count[c] = 0;
        __syncthreads();
        uint32_t rank = 0;
        if (valid) rank = atomicAdd(&s_count[cell], 1u);
        __syncthreads();
        for (uint32_t c = threadIdx.x; c < n_cells; c += 256) {
            const uint32_t k = s_count[c];
            if (k) s_base[c] = atomicAdd(&cursors[c * 32u], k);  // cursors 128 bytes apart
        }
        __syncthreads();
        if (valid) {
            const uint32_t at = s_base[cell] + rank;
            if (at < cell_capacity) {
                const size_t dst = static_cast<size_t>(cell) * cell_capacity + at;
                o[dst] = vo; d[dst] = vd; p[dst] = vp;
            }
        }
    }
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const uint32_t n_cells = argc > 2 ? atoi(argv[2]) : 64;
    const uint32_t n = argc > 3 ? strtoul(argv[3], nullptr, 10) : 100000000u;
    const int coherent = argc > 4 ? 1 : 0;
    // every cell's region holds four times its fair share (coherent blocks make some cells that much fuller)
    const uint32_t cell_capacity = mode ? static_cast<uint32_t>((4ull * n + n_cells - 1) / n_cells) : n;
    const size_t slots = mode ? static_cast<size_t>(cell_capacity) * n_cells : n;
    float4 *o, *d, *p;
    uint32_t* cursors;
    if (hipMalloc(&o, slots * 16) != hipSuccess || hipMalloc(&d, slots * 16) != hipSuccess || hipMalloc(&p, slots * 16) != hipSuccess ||
        hipMalloc(&cursors, static_cast<size_t>(n_cells) * 128 + 128) != hipSuccess) {
        printf("allocation failed\n");
        return 1;
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 5; ++it) {
        hipMemset(cursors, 0, static_cast<size_t>(n_cells) * 128 + 128);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const dim3 grid((n + 255) / 256), block(256);
        if (mode == 0) hipLaunchKernelGGL(append<0>, grid, block, 0, 0, o, d, p, n, n_cells, cell_capacity, cursors, coherent);
        else hipLaunchKernelGGL(append<1>, grid, block, 2 * n_cells * sizeof(uint32_t), 0, o, d, p, n, n_cells, cell_capacity, cursors, coherent);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    printf("mode %d cells %u rays %u %s: %.3f ms  (%.1f GB/s of records)\n", mode, n_cells, n, coherent ? "coherent" : "random", best,
           n * 48.0 / best / 1e6);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
