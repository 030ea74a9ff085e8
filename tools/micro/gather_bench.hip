// Microbenchmark: rate of divergent 16-byte gathers through the per-CU vector L1 on gfx950.
//   mode 0: one dwordx4 per lane from a random 64-byte record
//   mode 1: four dwordx4 per lane covering one random 64-byte record (the traversal kernel's node fetch)
//   mode 3: as mode 2 with the quad's record indices derived without a cross-lane shuffle (pure access-pattern rate)
//   mode 2: quad-cooperative: the 4 lanes of a quad fetch the 4 quarters of one record per instruction, 4 instructions
//           for the quad's 4 records (same bytes per lane as mode 1, coalesced inside the quad)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void gather(const float4* __restrict__ table, uint32_t n_records, int iters, float* out) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    uint32_t state = mix(gid + 1);
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        state = mix(state + it);
        uint32_t rec = state % n_records;
        if (MODE == 0) {
            const float4 v = table[static_cast<size_t>(rec) * 4 + (state >> 30)];
            acc += v.x + v.w;
        } else if (MODE == 1) {
            const float4* p = table + static_cast<size_t>(rec) * 4;
            const float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
        } else if (MODE == 3) {
            const uint32_t q = threadIdx.x & 3u;
            uint32_t qs = mix((gid >> 2) * 977u + it);  // the same in the 4 lanes of a quad
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                qs = mix(qs + k);
                const float4 v = table[static_cast<size_t>(qs % n_records) * 4 + q];
                acc += v.x + v.w;
            }
        } else {
            const uint32_t q = threadIdx.x & 3u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t rk = __shfl(rec, (threadIdx.x & ~3u) | k, 64);  // record of quad lane k
                const float4 v = table[static_cast<size_t>(rk) * 4 + q];
                acc += v.x + v.w;
            }
        }
        state ^= __float_as_uint(acc) & 1u;  // dependent chain like a traversal
    }
    out[gid] = acc;
}

int main(int argc, char** argv) {
    int dev = 0;
    hipSetDevice(dev);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, dev);
    const int cus = prop.multiProcessorCount;
    const double sizes_mb[] = {0.015625, 2.0, 64.0, 1024.0};
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 8 * cus);
    for (double mb : sizes_mb) {
        const uint32_t n_records = static_cast<uint32_t>(mb * 1024 * 1024 / 64);
        float4* table;
        hipMalloc(&table, static_cast<size_t>(n_records) * 64);
        hipMemset(table, 0, static_cast<size_t>(n_records) * 64);
        for (int mode = 0; mode < 4; ++mode) {
            for (int blocks_per_cu : {4, 8}) {
                const int blocks = blocks_per_cu * cus, iters = 2000;
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL(gather<0>, dim3(blocks), dim3(256), 0, 0, table, n_records, iters, out);
                    if (mode == 1) hipLaunchKernelGGL(gather<1>, dim3(blocks), dim3(256), 0, 0, table, n_records, iters, out);
                    if (mode == 3) hipLaunchKernelGGL(gather<3>, dim3(blocks), dim3(256), 0, 0, table, n_records, iters, out);
                    if (mode == 2) hipLaunchKernelGGL(gather<2>, dim3(blocks), dim3(256), 0, 0, table, n_records, iters, out);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                }
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                const double lane_loads = static_cast<double>(blocks) * 256 * iters * (mode == 0 ? 1 : 4);
                const double gbs = lane_loads * 16 / (ms * 1e-3) / 1e9;
                printf("table %8.3f MB mode %d waves/SIMD %d: %7.3f ms  %8.1f GB/s  %.3f lane-loads/clk/CU @2.4GHz\n", mb, mode,
                       blocks_per_cu, ms, gbs, lane_loads / (ms * 1e-3) / 2.4e9 / cus);
            }
        }
        hipFree(table);
    }
    return 0;
}
