// stream_probe4: the fused short-row kernel's streaming geometry with a variable ring depth -- one workgroup per CU, NS streaming
// waves that draw one row at a time from a cursor in LDS, K slots of 4 KiB (2 KiB of calls + 2 KiB of phreds, four LDS-DMA pieces)
// in flight per wave across row boundaries, counted s_waitcnt.  Question (round 5): is 96 KB in flight per CU (NS = 8, K = 3) what
// holds the stream at 0.73-0.75 of the HBM peak on 10 KB rows, i.e. would a fourth slot pay for the LDS it needs?
//   hipcc -O3 --offload-arch=gfx950 tools/probes/stream_probe4.hip -o tools/probes/bin/stream_probe4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ void glds4(uint32_t d0, const uint8_t *p0, uint32_t v0, const uint8_t *p1, uint32_t v1) {
    uint32_t keep, t;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[d0]\n\ts_add_u32 %[t], %[d0], 0x400\n\t"
        "global_load_lds_dwordx4 %[v0], %[p0] nt\n\ts_mov_b32 m0, %[t]\n\ts_add_u32 %[t], %[d0], 0x800\n\t"
        "global_load_lds_dwordx4 %[v1], %[p0] nt\n\ts_mov_b32 m0, %[t]\n\ts_add_u32 %[t], %[d0], 0xc00\n\t"
        "global_load_lds_dwordx4 %[v0], %[p1] nt\n\ts_mov_b32 m0, %[t]\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %[v1], %[p1] nt\n\ts_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep), [t] "=&s"(t) : [d0] "s"(d0), [p0] "s"(p0), [p1] "s"(p1), [v0] "v"(v0), [v1] "v"(v1) : "memory", "scc");
}
__device__ __forceinline__ const uint8_t *uni(const uint8_t *p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (const uint8_t *)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
template <int NS, int K, int NIDLE>
__global__ __launch_bounds__(64 * (NS + NIDLE)) void probe(const uint8_t *bs, const uint8_t *q, uint32_t n_sites, uint32_t n_samples, uint64_t pitch, uint32_t *sink) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];   // [NS][K][1024] ring, then the cursor
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *cursor = lds + NS * K * 1024;
    if (threadIdx.x == 0) { cursor[0] = 0; cursor[1] = 0; }
    __syncthreads();
    if (wave >= NS) {  // idle waves: what the solver waves are while nothing is queued
        for (uint32_t spins = 0; *(volatile uint32_t *)cursor < 0x40000000u && spins < (1u << 22); ++spins) __builtin_amdgcn_s_sleep(8);  // (bounded)
        return;
    }
    const uint32_t B0 = (uint32_t)((uint64_t)n_sites * blockIdx.x / gridDim.x), B1 = (uint32_t)((uint64_t)n_sites * (blockIdx.x + 1) / gridDim.x);
    const uint32_t n_chunks = (n_samples + 15u) >> 4, n_slots = (n_chunks + 127u) >> 7;
    const uint32_t ring_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_u32 *)(lds + wave * K * 1024));
    const uint32_t *ring = lds + wave * K * 1024;
    const uint32_t va = lane * 16u, vb = va + 1024u;
    const uint8_t *p0 = bs, *p1 = q;
    uint32_t p_left = 0, ring_w = 0, ring_r = 0, inflight = 0, rows = 0;
    bool done = false;
    auto issue = [&]() {
        if (p_left == 0u) {
            if (done) return;
            uint32_t c = 0;
            if (lane == 0) c = atomicAdd(cursor, 1u);
            c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
            if (c >= B1 - B0) { done = true; return; }
            const uint64_t off = (uint64_t)(B0 + c) * pitch;
            p0 = uni(bs + off); p1 = uni(q + off); p_left = n_slots; ++rows;
        }
        glds4(ring_lds + ring_w * 4096u, p0, va, p1, vb);   // (the last slot of a row reads a little past it: the planes are padded)
        p0 += 2048; p1 += 2048;
        --p_left;
        ring_w = (ring_w + 1u == (uint32_t)K) ? 0u : ring_w + 1u;
        ++inflight;
    };
#pragma unroll 1
    for (int k = 0; k < K; ++k) issue();
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll 1
    while (inflight) {
        if (inflight == (uint32_t)K) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (K - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t *rs = ring + ring_r * 1024 + lane * 4;
        acc ^= *reinterpret_cast<const u32x4 *>(rs) ^ *reinterpret_cast<const u32x4 *>(rs + 256) ^ *reinterpret_cast<const u32x4 *>(rs + 512) ^
               *reinterpret_cast<const u32x4 *>(rs + 768);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ring_r = (ring_r + 1u == (uint32_t)K) ? 0u : ring_r + 1u;
        --inflight;
        issue();
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = rows;
    if (NIDLE) { __builtin_amdgcn_s_waitcnt(0); if (lane == 0 && atomicAdd(cursor + 1, 1u) + 1u == (uint32_t)NS) *cursor = 0x40000000u; }
}
template <int NS, int K, int NIDLE>
double run(const uint8_t *a, const uint8_t *b, uint32_t S, uint32_t n, uint64_t pitch, uint32_t *sink) {
    const size_t dyn = (size_t)NS * K * 4096 + 64;
    hipFuncSetAttribute((const void *)probe<NS, K, NIDLE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NS, K, NIDLE><<<256, 64 * (NS + NIDLE), dyn>>>(a, b, S, n, pitch, sink);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) probe<NS, K, NIDLE><<<256, 64 * (NS + NIDLE), dyn>>>(a, b, S, n, pitch, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) return -1;
    return 2.0 * S * n * 5 / (ms * 1e-3) / 1e9;
}
int main() {
    const uint32_t n = 10000, S = 100000;
    const uint64_t pitch = 10240;
    uint8_t *a, *b; uint32_t *sink;
    hipMalloc(&a, S * pitch + 8192); hipMalloc(&b, S * pitch + 8192); hipMalloc(&sink, 64);
    hipMemset(a, 8, S * pitch + 8192); hipMemset(b, 30, S * pitch + 8192);
    printf("100000 rows of 10000 B per plane, one workgroup per CU; GB/s of the two planes (8000 = peak)\n");
    printf("NS 8 K 2 (64 KB in flight / CU): %6.0f\n", run<8, 2, 0>(a, b, S, n, pitch, sink));
    printf("NS 8 K 3 (96 KB)               : %6.0f   with 4 idle waves: %6.0f\n", run<8, 3, 0>(a, b, S, n, pitch, sink), run<8, 3, 4>(a, b, S, n, pitch, sink));
    printf("NS 8 K 4 (128 KB)              : %6.0f   with 4 idle waves: %6.0f\n", run<8, 4, 0>(a, b, S, n, pitch, sink), run<8, 4, 4>(a, b, S, n, pitch, sink));
    printf("NS 7 K 4 (112 KB)              : %6.0f\n", run<7, 4, 0>(a, b, S, n, pitch, sink));
    printf("NS 6 K 5 (120 KB)              : %6.0f\n", run<6, 5, 0>(a, b, S, n, pitch, sink));
    printf("NS 6 K 4 (96 KB)               : %6.0f\n", run<6, 4, 0>(a, b, S, n, pitch, sink));
    printf("NS 12 K 3 (144 KB)             : %6.0f\n", run<12, 3, 0>(a, b, S, n, pitch, sink));
    printf("NS 12 K 2 (96 KB)              : %6.0f\n", run<12, 2, 0>(a, b, S, n, pitch, sink));
    printf("NS 16 K 2 (128 KB)             : %6.0f\n", run<16, 2, 0>(a, b, S, n, pitch, sink));
    return 0;
}
