// stream_probe: how much of the HBM peak does a do-nothing stream of two planes of short rows reach as a function of the bytes
// each wave keeps in flight (VGPR loads, D x 1 KiB per wave) and of the waves per CU?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int D>
__global__ __launch_bounds__(256) void probe(const uint8_t *a, const uint8_t *b, size_t bytes_per_plane, uint32_t *sink) {
    const int lane = threadIdx.x & 63;
    const size_t n_waves = (size_t)gridDim.x * 4, gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t kib = bytes_per_plane >> 10;                 // 1 KiB granules per plane
    const size_t g0 = kib * gw / n_waves, g1 = kib * (gw + 1) / n_waves;
    u32x4 acc = {0, 0, 0, 0};
    constexpr int H = D / 2;                                  // granules per plane per step
    for (size_t g = g0; g + H <= g1; g += H) {
        u32x4 va[H], vb[H];
#pragma unroll
        for (int i = 0; i < H; ++i) {
            va[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a + ((g + i) << 10)) + lane);
            vb[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(b + ((g + i) << 10)) + lane);
        }
#pragma unroll
        for (int i = 0; i < H; ++i) acc ^= va[i] + vb[i];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}
template <int D>
double run(const uint8_t *a, const uint8_t *b, size_t bytes, uint32_t *sink, int wg_per_cu) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    probe<D><<<grid, 256>>>(a, b, bytes, sink);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) probe<D><<<grid, 256>>>(a, b, bytes, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return 2.0 * bytes * 5 / (ms * 1e-3) / 1e9;
}
int main() {
    const size_t bytes = (size_t)2 << 30;  // 2 GiB per plane
    uint8_t *a, *b; uint32_t *sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    for (int w : {2, 3, 4, 6, 8}) {
        printf("waves/CU %2d:", w * 4);
        printf("  D=4 %6.0f", run<4>(a, b, bytes, sink, w));
        printf("  D=8 %6.0f", run<8>(a, b, bytes, sink, w));
        printf("  D=16 %6.0f", run<16>(a, b, bytes, sink, w));
        printf("  D=32 %6.0f", run<32>(a, b, bytes, sink, w));
        printf("  GB/s\n");
    }
    return 0;
}
