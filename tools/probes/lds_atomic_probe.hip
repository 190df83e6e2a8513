// build: hipcc --offload-arch=gfx950 -O3 tools/probe/lds_atomic_probe.hip -o /tmp/lds_atomic_probe  (run on the GPU box)
// micro-benchmark: LDS instruction throughput per CU for the tally's access pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint32_t *idx, uint32_t *out, int iters, int active_mod) {
    __shared__ uint32_t hist[4][2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = lane; i < 2048; i += 64) hist[wave][i] = 0;
    __syncthreads();
    uint32_t x[16];
    for (int j = 0; j < 16; ++j) x[j] = idx[(blockIdx.x * 256 + threadIdx.x) * 16 + j];
    uint32_t *h = hist[wave];
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            uint32_t v = x[j];
            bool act = (v >> 16) < (uint32_t)active_mod;  // per-cell activity drawn on the host
            uint32_t a = v & 2047u;
            if (MODE == 0) { if (act) atomicAdd(&h[a], 1u); }
            if (MODE == 1) { if (act) h[a] = v; }
            if (MODE == 2) { if (act) acc += atomicAdd(&h[a], 1u); }
            if (MODE == 3) { if (act) acc += h[a]; }
            x[j] = v * 1664525u + 1013904223u;
        }
    }
    __syncthreads();
    uint32_t s = acc;
    for (int i = lane; i < 2048; i += 64) s += h[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    const int gridmax = 256 * 4, iters = 2000;
    const int grid = gridmax;
    uint32_t *idx, *out;
    CHECK(hipMalloc(&idx, grid * 256 * 16 * 4)); CHECK(hipMalloc(&out, grid * 256 * 4));
    uint32_t *h = (uint32_t *)malloc(grid * 256 * 16 * 4);
    for (int i = 0; i < grid * 256 * 16; ++i) h[i] = (uint32_t)rand() * 2654435761u + (uint32_t)rand();
    CHECK(hipMemcpy(idx, h, grid * 256 * 16 * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char *names[4] = {"ds_add_u32 (no rtn)", "ds_write_b32", "ds_add_rtn_u32", "ds_read_b32"};
    const int mods[4] = {65536, 5243 /* 8 % */, 1311 /* 2 % */, 0};
    for (int wg = 1; wg <= 4; wg *= 2)
    for (int mode = 0; mode < 2; ++mode)
        for (int m = 0; m < 4; ++m) {
            float ms = 0;
            const int g = 256 * wg;
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) k<0><<<g, 256>>>(idx, out, iters, mods[m]);
                if (mode == 1) k<1><<<g, 256>>>(idx, out, iters, mods[m]);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            // per CU: 16 waves x iters x 16 instructions
            double instr_per_cu = 4.0 * wg * iters * 16;
            printf("%2d waves/CU %-20s active %5.1f %%: %.3f ms  -> %.1f ns = %.1f cycles(2.4GHz) per wave-instruction per CU\n", 4 * wg, names[mode],
                   100.0 * mods[m] / 65536, ms, ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4);
        }
    return 0;
}
