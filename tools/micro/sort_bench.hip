// Microbenchmark: what ordering the rays of one launch by origin cell would cost on gfx950.
//   (1) hipcub radix sort of n (key, index) pairs, keys of `bits` significant bits;
//   (2) moving n 48-byte ray records (three float4 arrays) into the sorted order.
// usage: sort_bench [n = 73600000] [bits = 18]
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                              \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
__global__ __launch_bounds__(256) void fill(uint32_t* keys, uint32_t* idx, uint32_t n, uint32_t mask) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    keys[i] = mix(i + 1) & mask;
    idx[i] = i;
}
__global__ __launch_bounds__(256) void move(const float4* __restrict__ o, const float4* __restrict__ d, const float4* __restrict__ p,
                                            const uint32_t* __restrict__ perm, uint32_t n, float4* __restrict__ o2,
                                            float4* __restrict__ d2, float4* __restrict__ p2) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = perm[i];
    o2[i] = o[s];
    d2[i] = d[s];
    p2[i] = p[s];
}

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? static_cast<uint32_t>(std::atoll(argv[1])) : 73600000u;
    const int bits = argc > 2 ? std::atoi(argv[2]) : 18;
    uint32_t *k0, *k1, *v0, *v1;
    CHECK(hipMalloc(&k0, n * 4ull));
    CHECK(hipMalloc(&k1, n * 4ull));
    CHECK(hipMalloc(&v0, n * 4ull));
    CHECK(hipMalloc(&v1, n * 4ull));
    float4* rec[6];
    for (auto& r : rec) CHECK(hipMalloc(&r, n * 16ull));
    for (int i = 0; i < 3; ++i) CHECK(hipMemset(rec[i], 0, n * 16ull));
    size_t tmp_bytes = 0;
    CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k0, k1, v0, v1, static_cast<int>(n), 0, bits));
    void* tmp;
    CHECK(hipMalloc(&tmp, tmp_bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const dim3 grid((n + 255) / 256), block(256);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(fill, grid, block, 0, 0, k0, v0, n, (bits >= 32) ? 0xffffffffu : ((1u << bits) - 1u));
        CHECK(hipEventRecord(e0));
        CHECK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, k0, k1, v0, v1, static_cast<int>(n), 0, bits));
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms_sort = 0;
        CHECK(hipEventElapsedTime(&ms_sort, e0, e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(move, grid, block, 0, 0, rec[0], rec[1], rec[2], v1, n, rec[3], rec[4], rec[5]);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms_move = 0;
        CHECK(hipEventElapsedTime(&ms_move, e0, e1));
        std::printf("n %u bits %d: sort pairs %.3f ms, move 48-byte records %.3f ms (temporary storage %.1f MB)\n", n, bits, ms_sort, ms_move,
                    tmp_bytes / 1e6);
    }
    CHECK(hipGetLastError());
    return 0;
}
