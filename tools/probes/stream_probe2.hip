// stream_probe2: the same do-nothing stream of two planes, (B) through a per-wave LDS-DMA ring of K slots of U KiB per plane
// shaped like bv_p1s_stream_kernel (wait for the oldest slot, read it from LDS, refill it), (C) with plain loads but walking
// ROWS of n bytes at a pitch (a row's last granule partial), double-buffered.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ void glds16(uint32_t lds_dst, const uint8_t *base, uint32_t voff) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(base) : "memory");
}
__device__ __forceinline__ const uint8_t *uni(const uint8_t *p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (const uint8_t *)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
template <int K, int U, bool READ>
__global__ __launch_bounds__(256) void probe_dma(const uint8_t *a, const uint8_t *b, size_t bytes_per_plane, uint32_t *sink) {
    __shared__ __attribute__((aligned(16))) uint32_t ring[4][K][U * 512];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const size_t n_waves = (size_t)gridDim.x * 4, gw = (size_t)blockIdx.x * 4 + wave;
    const size_t slots = bytes_per_plane / (1024u * U);
    const size_t s0 = slots * gw / n_waves, s1 = slots * (gw + 1) / n_waves;
    const uint32_t ring_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_u32 *)ring[wave][0]);
    const uint32_t voff = lane * 16u;
    size_t p = s0;
    uint32_t ring_w = 0, ring_r = 0, inflight = 0;
    auto issue = [&]() {
        if (p < s1) {
            const uint8_t *pa = uni(a + p * 1024u * U), *pb = uni(b + p * 1024u * U);
            const uint32_t dst = ring_lds + ring_w * (U * 2048u);
#pragma unroll
            for (int u = 0; u < U; ++u) { glds16(dst + 1024u * u, pa + 1024u * u, voff); glds16(dst + 1024u * (U + u), pb + 1024u * u, voff); }
            ring_w = (ring_w + 1 == K) ? 0 : ring_w + 1; ++inflight; ++p;
        }
    };
    for (int k = 0; k < K; ++k) issue();
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll 1
    for (size_t s = s0; s < s1; ++s) {
        if (inflight == (uint32_t)K) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * U * (K - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (READ) {
#pragma unroll
            for (int u = 0; u < 2 * U; ++u) acc ^= *reinterpret_cast<const u32x4 *>(&ring[wave][ring_r][u * 256 + lane * 4]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        ring_r = (ring_r + 1 == K) ? 0 : ring_r + 1; --inflight;
        issue();
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}
// rows of n bytes at pitch P, each wave a contiguous range of rows; per row ceil(n / 1024) granules per plane, the last one partial
__global__ __launch_bounds__(256) void probe_rows(const uint8_t *a, const uint8_t *b, uint32_t n_rows, uint32_t n, uint32_t P, uint32_t *sink) {
    const int lane = threadIdx.x & 63;
    const size_t n_waves = (size_t)gridDim.x * 4, gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t r0 = (uint32_t)((size_t)n_rows * gw / n_waves), r1 = (uint32_t)((size_t)n_rows * (gw + 1) / n_waves);
    const uint32_t n_chunks = (n + 15u) >> 4;
    u32x4 acc = {0, 0, 0, 0};
    for (uint32_t r = r0; r < r1; ++r) {
        const u32x4 *pa = reinterpret_cast<const u32x4 *>(a + (size_t)r * P), *pb = reinterpret_cast<const u32x4 *>(b + (size_t)r * P);
        for (uint32_t c0 = 0; c0 < n_chunks; c0 += 128) {
            u32x4 va0 = {0, 0, 0, 0}, va1 = va0, vb0 = va0, vb1 = va0;
            const uint32_t c = c0 + lane;
            if (c < n_chunks) { va0 = __builtin_nontemporal_load(pa + c); vb0 = __builtin_nontemporal_load(pb + c); }
            if (c + 64 < n_chunks) { va1 = __builtin_nontemporal_load(pa + c + 64); vb1 = __builtin_nontemporal_load(pb + c + 64); }
            acc ^= va0 + vb0 + va1 + vb1;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}
__global__ void fill_random(uint32_t *p, size_t n, uint32_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed * 40503u;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = x;
    }
}
template <typename F>
double timeit(F f, double bytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return bytes * 5 / (ms * 1e-3) / 1e9;
}
int main() {
    const size_t bytes = (size_t)2 << 30;
    uint8_t *a, *b; uint32_t *sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) {
        fill_random<<<4096, 256>>>(reinterpret_cast<uint32_t *>(a), bytes / 4, 1u);
        fill_random<<<4096, 256>>>(reinterpret_cast<uint32_t *>(b), bytes / 4, 2u);
        hipDeviceSynchronize();
        printf("-- pseudo-random bytes instead of constant ones\n");
    }
    for (int w : {2, 3}) {
        printf("DMA ring, %d waves/CU:", w * 4);
        printf("  K3U2 read %6.0f", timeit([&] { probe_dma<3, 2, true><<<256 * w, 256>>>(a, b, bytes, sink); }, 2.0 * bytes));
        printf("  K3U2 noread %6.0f", timeit([&] { probe_dma<3, 2, false><<<256 * w, 256>>>(a, b, bytes, sink); }, 2.0 * bytes));
        printf("  K4U1 read %6.0f", timeit([&] { probe_dma<4, 1, true><<<256 * w, 256>>>(a, b, bytes, sink); }, 2.0 * bytes));
        printf("  K6U1 read %6.0f", timeit([&] { probe_dma<6, 1, true><<<256 * w, 256>>>(a, b, bytes, sink); }, 2.0 * bytes));
        printf("  GB/s\n");
    }
    for (uint32_t n : {10000u, 3000u, 40000u}) {
        const uint32_t P = (n + 255) / 256 * 256, rows = (uint32_t)(bytes / P);
        for (int w : {2, 3, 4})
            printf("rows of %u B (pitch %u), plain loads, %d waves/CU: %6.0f GB/s (row bytes only)\n", n, P, w * 4,
                   timeit([&] { probe_rows<<<256 * w, 256>>>(a, b, rows, n, P, sink); }, 2.0 * rows * (double)n));
    }
    }
    return 0;
}
