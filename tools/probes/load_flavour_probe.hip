// build: hipcc --offload-arch=gfx950 -O3 tools/probe/stream_probe.hip -o /tmp/stream_probe  (run on the GPU box)
// micro-benchmark: what read bandwidth does a streaming kernel get from the HBM, by load flavour (non-temporal or
// plain), 16-byte loads in flight per lane, and resident waves per CU?  Rows of 100 KB x 2 planes, ticket counter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k(const uint8_t *p0, const uint8_t *p1, size_t pitch, uint32_t rows, uint32_t row_bytes,
                                         uint32_t *ticket, uint32_t *out) {
    const int lane = threadIdx.x & 63;
    u4 acc = {0, 0, 0, 0};
    for (;;) {
        uint32_t r = 0;
        if (lane == 0) r = atomicAdd(ticket, 1u);
        r = __builtin_amdgcn_readfirstlane(r);
        if (r >= rows) break;
        const u4 *a = (const u4 *)(p0 + (size_t)r * pitch), *b = (const u4 *)(p1 + (size_t)r * pitch);
        const uint32_t n = row_bytes / 16;
        for (uint32_t i = lane; i < n; i += 64 * U) {
            u4 va[U], vb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                uint32_t j = i + u * 64;
                if (j < n) {
                    if (NT) { va[u] = __builtin_nontemporal_load(a + j); vb[u] = __builtin_nontemporal_load(b + j); }
                    else { va[u] = a[j]; vb[u] = b[j]; }
                } else { va[u] = u4{0,0,0,0}; vb[u] = u4{0,0,0,0}; }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= va[u] + vb[u];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[threadIdx.x] = 1;
}
template <int U, bool NT>
static void run(const uint8_t *p0, const uint8_t *p1, size_t total, uint32_t L, size_t pitch, uint32_t *ticket, uint32_t *out, int waves_per_cu) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const uint32_t rows = (uint32_t)(total / pitch);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(ticket, 0, 4));
        CHECK(hipEventRecord(e0));
        k<U, NT><<<256 * waves_per_cu / 4, 256>>>(p0, p1, pitch, rows, L, ticket, out);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("%2d waves/CU  %s  %2d x16B in flight per lane and plane: %.3f ms  %.0f GB/s\n", waves_per_cu, NT ? "nt   " : "plain", U, ms,
           2.0 * rows * (double)L / ms / 1e6);
}
int main() {
    const size_t total = (size_t)6 << 30;  // bytes per plane
    uint8_t *p0, *p1; uint32_t *ticket, *out;
    CHECK(hipMalloc(&p0, total)); CHECK(hipMalloc(&p1, total)); CHECK(hipMalloc(&ticket, 4)); CHECK(hipMalloc(&out, 1024));
    CHECK(hipMemset(p0, 1, total)); CHECK(hipMemset(p1, 2, total));
    const uint32_t L = 100000; const size_t pitch = 100096;
    for (int w : {4, 8, 12, 16, 24}) {
        run<2, true>(p0, p1, total, L, pitch, ticket, out, w);
        run<4, true>(p0, p1, total, L, pitch, ticket, out, w);
        run<8, true>(p0, p1, total, L, pitch, ticket, out, w);
        run<4, false>(p0, p1, total, L, pitch, ticket, out, w);
        run<8, false>(p0, p1, total, L, pitch, ticket, out, w);
    }
    return 0;
}
