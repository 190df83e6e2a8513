// stream_probe3: bv_p1s_stream_kernel's loop skeleton (rows, partial last slot, counted waits, LDS reads) without tally and
// epilogue, to bisect what separates it from the bare ring of stream_probe2 (7.0 TB/s).
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
// MODE bit 0: dummy loads for KiB past the row end (the kernel's rule); bit 1: a fake epilogue (16 LDS reads + reductions + store);
// bit 2: zero 4 KiB of LDS per row
template <int K, int U, int MODE>
__global__ __launch_bounds__(256) void probe(const uint8_t *bs, const uint8_t *q, uint32_t n_sites, uint32_t n_samples, uint64_t pitch, uint32_t *sink) {
    __shared__ __attribute__((aligned(16))) uint32_t ring[4][K][U * 512];
    __shared__ __attribute__((aligned(16))) uint32_t hist[4][1024 + 16];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t n_waves = (uint64_t)gridDim.x * 4, gw = (uint64_t)blockIdx.x * 4 + wave;
    const uint32_t s0 = (uint32_t)((uint64_t)n_sites * gw / n_waves), s1 = (uint32_t)((uint64_t)n_sites * (gw + 1) / n_waves);
    if (s0 >= s1) return;
    const uint32_t n_chunks = (n_samples + 15u) >> 4, n_slots = (n_chunks + 64u * U - 1u) / (64u * U);
    const uint32_t last_valid = n_chunks - (n_slots - 1u) * 64u * U;
    const uint32_t ring_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_u32 *)ring[wave][0]);
    const uint32_t voff = lane * 16u;
    uint32_t p_site = s0, p_j = 0, ring_w = 0, inflight = 0;
    auto issue = [&]() {
        if (p_site < s1) {
            const size_t off = (size_t)p_site * pitch + (size_t)p_j * (1024u * U);
            const uint8_t *pb = uni(bs + off), *pq = uni(q + off);
            const uint32_t dst = ring_lds + ring_w * (U * 2048u);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool any = p_j + 1u < n_slots || 64u * u < last_valid;
                if (!(MODE & 1) && !any) continue;
                const uint8_t *sb = any ? pb + 1024u * u : uni(bs + (size_t)p_site * pitch);
                const uint8_t *sq = any ? pq + 1024u * u : uni(q + (size_t)p_site * pitch);
                if (any ? (p_j + 1u < n_slots || (uint32_t)lane + 64u * u < last_valid) : lane == 0) {
                    glds16(dst + 1024u * u, sb, voff);
                    glds16(dst + 1024u * (U + u), sq, voff);
                }
            }
            ring_w = (ring_w + 1u == (uint32_t)K) ? 0u : ring_w + 1u;
            ++inflight;
            if (++p_j == n_slots) { p_j = 0; ++p_site; }
        }
    };
#pragma unroll 1
    for (int k = 0; k < K; ++k) issue();
    uint32_t ring_r = 0;
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll 1
    for (uint32_t site = s0; site < s1; ++site) {
#pragma unroll 1
        for (uint32_t j = 0; j < n_slots; ++j) {
            if (MODE & 1) {
                if (inflight == (uint32_t)K) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * U * (K - 1)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // without the dummy loads the count is not fixed
            }
#pragma unroll
            for (int u = 0; u < 2 * U; ++u) acc ^= *reinterpret_cast<const u32x4 *>(&ring[wave][ring_r][u * 256 + lane * 4]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ring_r = (ring_r + 1u == (uint32_t)K) ? 0u : ring_r + 1u;
            --inflight;
            issue();
        }
        if (MODE & 2) {
            uint32_t t = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) t += hist[wave][i * 64 + lane];
            for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
            if (lane < 12) sink[16 + (size_t)site * 12 + lane] = t + acc.x;
        }
        if (MODE & 4) {
            uint4 *h4 = reinterpret_cast<uint4 *>(hist[wave]);
#pragma unroll
            for (int i = 0; i < 4; ++i) h4[i * 64 + lane] = make_uint4(0, 0, 0, 0);
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
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
    const uint32_t n = 10000, S = 100000;
    const uint64_t pitch = 10240;
    uint8_t *a, *b; uint32_t *sink;
    hipMalloc(&a, S * pitch + 4096); hipMalloc(&b, S * pitch + 4096); hipMalloc(&sink, 64 + (size_t)S * 48);
    hipMemset(a, 8, S * pitch); hipMemset(b, 30, S * pitch);
    const double bytes = 2.0 * S * n;
    for (int w : {2, 3}) {
        printf("%d waves/CU, 100000 rows of 10000 B:", w * 4);
        printf("  bare(vmcnt0) %6.0f", timeit([&] { probe<3, 2, 0><<<256 * w, 256>>>(a, b, S, n, pitch, sink); }, bytes));
        printf("  counted %6.0f", timeit([&] { probe<3, 2, 1><<<256 * w, 256>>>(a, b, S, n, pitch, sink); }, bytes));
        printf("  +epilogue %6.0f", timeit([&] { probe<3, 2, 3><<<256 * w, 256>>>(a, b, S, n, pitch, sink); }, bytes));
        printf("  +zeroing %6.0f", timeit([&] { probe<3, 2, 7><<<256 * w, 256>>>(a, b, S, n, pitch, sink); }, bytes));
        printf("  K4U1 all %6.0f", timeit([&] { probe<4, 1, 7><<<256 * w, 256>>>(a, b, S, n, pitch, sink); }, bytes));
        printf("  GB/s\n");
    }
    const uint32_t S5 = 524288;
    uint8_t *a5, *b5; uint32_t *sink5;
    hipMalloc(&a5, S5 * pitch + 4096); hipMalloc(&b5, S5 * pitch + 4096); hipMalloc(&sink5, 64 + (size_t)S5 * 48);
    hipMemset(a5, 8, S5 * pitch); hipMemset(b5, 30, S5 * pitch);
    printf("8 waves/CU, 524288 rows: counted+epilogue+zeroing %6.0f GB/s\n",
           timeit([&] { probe<3, 2, 7><<<512, 256>>>(a5, b5, S5, n, pitch, sink5); }, 2.0 * S5 * n));
    return 0;
}
