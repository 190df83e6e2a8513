// build: hipcc --offload-arch=gfx950 -O3 tools/probe/rowread_probe.hip -o /tmp/rowread_probe  (run on the GPU box)
// micro-benchmark: HBM read bandwidth when every wave streams whole rows of L bytes from each of two planes
// (the short-row access pattern of pass 1), rows handed out by a ticket counter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
template <int NT_LOADS>
__global__ __launch_bounds__(256) void k(const uint8_t *p0, const uint8_t *p1, size_t pitch, uint32_t rows, uint32_t row_bytes,
                                         uint32_t *ticket, uint32_t *out, int chunk) {
    const int lane = threadIdx.x & 63;
    u4 acc = {0, 0, 0, 0};
    for (;;) {
        uint32_t r0 = 0;
        if (lane == 0) r0 = atomicAdd(ticket, (uint32_t)chunk);
        r0 = __builtin_amdgcn_readfirstlane(r0);
        if (r0 >= rows) break;
        for (uint32_t r = r0; r < r0 + chunk && r < rows; ++r) {
            const u4 *a = (const u4 *)(p0 + (size_t)r * pitch), *b = (const u4 *)(p1 + (size_t)r * pitch);
            const uint32_t n = row_bytes / 16;
            for (uint32_t i = lane; i < n; i += 64 * NT_LOADS) {
                u4 va[NT_LOADS], vb[NT_LOADS];
#pragma unroll
                for (int u = 0; u < NT_LOADS; ++u) {
                    uint32_t j = i + u * 64;
                    if (j < n) { va[u] = __builtin_nontemporal_load(a + j); vb[u] = __builtin_nontemporal_load(b + j); }
                    else { va[u] = u4{0,0,0,0}; vb[u] = u4{0,0,0,0}; }
                }
#pragma unroll
                for (int u = 0; u < NT_LOADS; ++u) acc ^= va[u] + vb[u];
            }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[threadIdx.x] = 1;
}
int main() {
    const size_t total = (size_t)3 << 30;  // bytes per plane
    uint8_t *p0, *p1; uint32_t *ticket, *out;
    CHECK(hipMalloc(&p0, total)); CHECK(hipMalloc(&p1, total)); CHECK(hipMalloc(&ticket, 4)); CHECK(hipMalloc(&out, 1024));
    CHECK(hipMemset(p0, 1, total)); CHECK(hipMemset(p1, 2, total));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const uint32_t lens[] = {2560, 10240, 30720, 102400, 1024000};
    for (int wg = 2; wg <= 8; wg *= 2)
    for (uint32_t L : lens) {
        const uint32_t rows = (uint32_t)(total / L);
        for (int chunk = 1; chunk <= 4; chunk *= 4) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipMemset(ticket, 0, 4));
                CHECK(hipEventRecord(e0));
                k<4><<<256 * wg, 256>>>(p0, p1, L, rows, L, ticket, out, chunk);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("%2d waves/CU  row %7u B x2 planes, ticket %d: %.3f ms  %.0f GB/s\n", 4 * wg, L, chunk, ms, 2.0 * rows * (double)L / ms / 1e6);
        }
    }
    return 0;
}
