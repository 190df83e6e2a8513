// stream_probe4: the fused short-row kernel's streaming geometry with a variable ring depth -- one workgroup per CU, NS streaming
// waves that draw one row at a time from a cursor in LDS, K slots of 4 KiB (2 KiB of calls + 2 KiB of phreds, four LDS-DMA pieces)
// in flight per wave across row boundaries, counted s_waitcnt.  Question (round 5): is 96 KB in flight per CU (NS = 8, K = 3) what
// holds the stream at 0.73-0.75 of the HBM peak on 10 KB rows, i.e. would a fourth slot pay for the LDS it needs?
//   hipcc -O3 --offload-arch=gfx950 tools/probes/stream_probe4.hip -o tools/probes/bin/stream_probe4
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../basevar_amd/csrc/bv_tally.h"   // the kernel's own per-cell tally (MODE bit 1)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) uint32_t lds_u32;
// a row's last slot: lanes past the row's end load nothing (the kernel's bv_f_glds4_masked)
__device__ __forceinline__ void glds4_masked(uint32_t d0, const uint8_t *p0, uint32_t v0, uint32_t v1, const uint8_t *p1, unsigned long long mA, unsigned long long mB) {
    uint32_t keep, t;
    unsigned long long sv;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\ts_mov_b64 %[sv], exec\n\ts_mov_b64 exec, %[mA]\n\ts_mov_b32 m0, %[d0]\n\ts_add_u32 %[t], %[d0], 0x800\n\t"
        "global_load_lds_dwordx4 %[v0], %[p0] nt\n\ts_mov_b32 m0, %[t]\n\ts_add_u32 %[t], %[d0], 0x400\n\t"
        "global_load_lds_dwordx4 %[v0], %[p1] nt\n\ts_mov_b64 exec, %[mB]\n\ts_mov_b32 m0, %[t]\n\ts_add_u32 %[t], %[d0], 0xc00\n\t"
        "global_load_lds_dwordx4 %[v1], %[p0] nt\n\ts_mov_b32 m0, %[t]\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %[v1], %[p1] nt\n\ts_mov_b64 exec, %[sv]\n\ts_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep), [t] "=&s"(t), [sv] "=&s"(sv) : [d0] "s"(d0), [p0] "s"(p0), [p1] "s"(p1), [v0] "v"(v0), [v1] "v"(v1), [mA] "s"(mA), [mB] "s"(mB) : "memory", "scc");
}
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
// three planes, the pass-2 rows' slot: 1 KiB of a, 1 KiB of b, 2 KiB of c
__device__ __forceinline__ void glds4_p2(uint32_t d0, const uint8_t *pa, const uint8_t *pb, const uint8_t *pc, uint32_t v0, uint32_t v1) {
    uint32_t keep, t;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[d0]\n\ts_add_u32 %[t], %[d0], 0x400\n\t"
        "global_load_lds_dwordx4 %[v0], %[pa] nt\n\ts_mov_b32 m0, %[t]\n\ts_add_u32 %[t], %[d0], 0x800\n\t"
        "global_load_lds_dwordx4 %[v0], %[pb] nt\n\ts_mov_b32 m0, %[t]\n\ts_add_u32 %[t], %[d0], 0xc00\n\t"
        "global_load_lds_dwordx4 %[v0], %[pc] nt\n\ts_mov_b32 m0, %[t]\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %[v1], %[pc] nt\n\ts_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep), [t] "=&s"(t) : [d0] "s"(d0), [pa] "s"(pa), [pb] "s"(pb), [pc] "s"(pc), [v0] "v"(v0), [v1] "v"(v1) : "memory", "scc");
}
// MODE bit 4 (16): the pass-2 rows' geometry -- slots of 1,024 cells: 1 KiB of plane a, 1 KiB of plane b, 2 KiB of the 2-byte
// plane c; bit 5 (32): the same bytes as slots of 2,048 cells that alternate between (2 KiB of a + 2 KiB of b) and (4 KiB of c)
// MODE bit 0: rows end where they end (masked last slot, no read past the row); bit 1: the kernel's per-cell tally into an 8 x 128
// histogram; bit 2: a per-row epilogue (16 LDS reads, wave reductions, a 48-byte store); bit 3: 4 KiB of LDS zeroed per row
template <int NS, int K, int NIDLE, int MODE = 0>
__global__ __launch_bounds__(64 * (NS + NIDLE)) void probe(const uint8_t *bs, const uint8_t *q, uint32_t n_sites, uint32_t n_samples, uint64_t pitch, uint32_t *sink, const uint8_t *r16 = nullptr) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];   // [NS][K][1024] ring, the cursor (16 words), [NS][1040] histograms
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *cursor = lds + NS * K * 1024;
    if (threadIdx.x == 0) { cursor[0] = 0; cursor[1] = 0; }
    __syncthreads();
    if (wave >= NS) {  // idle waves: what the solver waves are while nothing is queued
        for (uint32_t spins = 0; *(volatile uint32_t *)cursor < 0x40000000u && spins < (1u << 22); ++spins) __builtin_amdgcn_s_sleep(8);  // (bounded)
        return;
    }
    const uint32_t B0 = (uint32_t)((uint64_t)n_sites * blockIdx.x / gridDim.x), B1 = (uint32_t)((uint64_t)n_sites * (blockIdx.x + 1) / gridDim.x);
    const uint32_t n_chunks = (n_samples + 15u) >> 4, n_slots = (MODE & 16) ? (n_chunks + 63u) >> 6 : ((MODE & 32) ? 2u * ((n_chunks + 127u) >> 7) : (n_chunks + 127u) >> 7);
    const uint32_t ring_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_u32 *)(lds + wave * K * 1024));
    const uint32_t *ring = lds + wave * K * 1024;
    const uint32_t va = lane * 16u, vb = va + 1024u;
    uint32_t *hist = lds + NS * K * 1024 + 16 + wave * 1040;
    if (MODE & 6) { for (int i = lane; i < 1040; i += 64) hist[i] = 0; }
    const uint32_t last1 = n_chunks - (n_slots - 1u) * 128u;
    const uint32_t vbl = last1 > 64u ? vb : va;
    const unsigned long long mA1 = last1 >= 64u ? ~0ull : ((1ull << last1) - 1ull);
    const unsigned long long mB1 = last1 > 64u ? (last1 >= 128u ? ~0ull : ((1ull << (last1 - 64u)) - 1ull)) : 1ull;
    uint32_t one;
    asm volatile("v_mov_b32 %0, 1" : "=v"(one));
    const uint8_t *p0 = bs, *p1 = q, *p2 = r16;
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
            if (MODE & 48) p2 = uni(r16 + 2u * off);
        }
        if (MODE & 16) { glds4_p2(ring_lds + ring_w * 4096u, uni(p0), uni(p1), uni(p2), va, vb); p0 += 1024; p1 += 1024; p2 += 2048; }
        else if (MODE & 32) {
            if (p_left & 1u) { glds4(ring_lds + ring_w * 4096u, uni(p2), va, uni(p2 + 2048), vb); p2 += 4096; }   // (second of a pair: the ranks)
            else { glds4(ring_lds + ring_w * 4096u, uni(p0), va, uni(p1), vb); p0 += 2048; p1 += 2048; }
        } else {
        if (!(MODE & 1) || p_left > 1u) glds4(ring_lds + ring_w * 4096u, p0, va, p1, vb);   // (MODE bit 0 clear: the last slot reads a little past the row)
        else glds4_masked(ring_lds + ring_w * 4096u, p0, va, vbl, p1, mA1, mB1);
        p0 += 2048; p1 += 2048;
        }
        --p_left;
        ring_w = (ring_w + 1u == (uint32_t)K) ? 0u : ring_w + 1u;
        ++inflight;
    };
#pragma unroll 1
    for (int k = 0; k < K; ++k) issue();
    u32x4 acc = {0, 0, 0, 0};
    uint32_t j = 0, stores = 0, row_id = 0;   // slot within the row being consumed; stores of the previous row still allowed for
#pragma unroll 1
    while (inflight) {
        if (inflight != (uint32_t)K) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (stores && j < 3u) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (K - 1) + 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (K - 1)) : "memory");
        const uint32_t *rs = ring + ring_r * 1024 + lane * 4;
        bv_u32x4 w0 = *reinterpret_cast<const bv_u32x4 *>(rs), w1 = *reinterpret_cast<const bv_u32x4 *>(rs + 256);
        bv_u32x4 w2 = *reinterpret_cast<const bv_u32x4 *>(rs + 512), w3 = *reinterpret_cast<const bv_u32x4 *>(rs + 768);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ring_r = (ring_r + 1u == (uint32_t)K) ? 0u : ring_r + 1u;
        --inflight;
        issue();
        if (MODE & 2) {
            w2.x <<= 1; w2.y <<= 1; w2.z <<= 1; w2.w <<= 1; w3.x <<= 1; w3.y <<= 1; w3.z <<= 1; w3.w <<= 1;
            bv_tally_chunk<1>(w0, w2, hist, one);
            bv_tally_chunk<1>(w1, w3, hist, one);
        } else {
            acc ^= (u32x4){w0.x, w0.y, w0.z, w0.w} ^ (u32x4){w1.x, w1.y, w1.z, w1.w} ^ (u32x4){w2.x, w2.y, w2.z, w2.w} ^ (u32x4){w3.x, w3.y, w3.z, w3.w};
        }
        if (++j == n_slots) {  // end of a row
            j = 0; stores = 0;
            if (MODE & 4) {
                uint32_t t = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) t += hist[i * 64 + lane];
                for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
                if (lane < 12) sink[16 + ((size_t)blockIdx.x * 4096 + (row_id & 4095u)) * 12 + lane] = t + acc.x;
                stores = 1;
            }
            if (MODE & 8) {
                uint4 *h4 = reinterpret_cast<uint4 *>(hist);
#pragma unroll
                for (int i = 0; i < 4; ++i) h4[i * 64 + lane] = make_uint4(0, 0, 0, 0);
            }
            ++row_id;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = rows;
    if (NIDLE) { __builtin_amdgcn_s_waitcnt(0); if (lane == 0 && atomicAdd(cursor + 1, 1u) + 1u == (uint32_t)NS) *cursor = 0x40000000u; }
}
template <int NS, int K, int NIDLE, int MODE = 0>
double run(const uint8_t *a, const uint8_t *b, uint32_t S, uint32_t n, uint64_t pitch, uint32_t *sink, const uint8_t *c = nullptr) {
    const size_t dyn = (size_t)NS * K * 4096 + 64 + (size_t)NS * 1040 * 4;
    hipFuncSetAttribute((const void *)probe<NS, K, NIDLE, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NS, K, NIDLE, MODE><<<256, 64 * (NS + NIDLE), dyn>>>(a, b, S, n, pitch, sink, c);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) probe<NS, K, NIDLE, MODE><<<256, 64 * (NS + NIDLE), dyn>>>(a, b, S, n, pitch, sink, c);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) return -1;
    return ((MODE & 48) ? 4.0 : 2.0) * S * n * 5 / (ms * 1e-3) / 1e9;
}
int main() {
    const uint32_t n = 10000, S = 100000;
    const uint64_t pitch = 10240;
    uint8_t *a, *b; uint32_t *sink;
    hipMalloc(&a, S * pitch + 8192); hipMalloc(&b, S * pitch + 8192); hipMalloc(&sink, 64 + (size_t)256 * 4096 * 48 + 4096);
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
    printf("NS 8 K 3 + 4 idle waves, layers of the kernel's streaming loop (uncovered rows: the tally's adds are all masked off):\n");
    printf("  exact rows (masked last slot)          : %6.0f\n", run<8, 3, 4, 1>(a, b, S, n, pitch, sink));
    printf("  + tally                                : %6.0f\n", run<8, 3, 4, 3>(a, b, S, n, pitch, sink));
    printf("  + tally + epilogue (reads, sums, store): %6.0f\n", run<8, 3, 4, 7>(a, b, S, n, pitch, sink));
    printf("  + tally + epilogue + zeroing           : %6.0f\n", run<8, 3, 4, 15>(a, b, S, n, pitch, sink));
    printf("  exact rows + epilogue + zeroing        : %6.0f\n", run<8, 3, 4, 13>(a, b, S, n, pitch, sink));
    {
        uint8_t *c;
        hipMalloc(&c, 2 * S * pitch + 16384); hipMemset(c, 1, 2 * S * pitch + 16384);
        printf("pass-2 geometry (three planes, 4 B per cell; GB/s of all three), NS 8 K 3 + 4 idle waves, bare:\n");
        printf("  slots of 1 KiB + 1 KiB + 2 KiB (the kernel's)        : %6.0f\n", run<8, 3, 4, 16>(a, b, S, n, pitch, sink, c));
        printf("  slots alternating (2 KiB + 2 KiB) / (4 KiB)           : %6.0f\n", run<8, 3, 4, 32>(a, b, S, n, pitch, sink, c));
        printf("  the pass-1 geometry again (two planes, 2 KiB + 2 KiB) : %6.0f\n", run<8, 3, 4, 0>(a, b, S, n, pitch, sink));
        hipFree(c);
    }
    hipMemset(a, 2, S * pitch + 8192);   // every cell covered: the worst case for the adds
    printf("  all layers, every cell a covered 'G'   : %6.0f\n", run<8, 3, 4, 15>(a, b, S, n, pitch, sink));
    return 0;
}
