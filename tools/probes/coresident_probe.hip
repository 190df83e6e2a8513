// coresident_probe.hip -- can a small workgroup be placed on a CU beside two big-LDS workgroups of a persistent kernel?
// Kernel A: 2 workgroups per CU, 256 threads, LDS_A bytes of LDS, ~80 VGPRs, spins for 300 us.  20 us after its launch, kernel B
// (256 workgroups x 128 or 256 threads, X bytes of LDS, V VGPRs, scratch or not) is launched on another stream; it does
// nothing.  Its duration tells whether its workgroups had to wait for A's to leave.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>

__global__ __launch_bounds__(256) void hog(unsigned long long ticks, unsigned *sink) {
    extern __shared__ unsigned lds[];
    asm volatile("v_mov_b32 v76, 0" ::: "v76");  // ~80 VGPRs, as the streaming kernel
    lds[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (lds[(threadIdx.x + 1) & 255] == 12345u) atomicAdd(sink, 1u);
}
// a streaming hog: every wave reads its own contiguous range of a big buffer with 16-byte loads, 8 in flight per lane
__global__ __launch_bounds__(256) void stream_hog(const uint4 *buf, size_t n16_per_wave, int reps, unsigned *sink) {
    extern __shared__ unsigned lds[];
    asm volatile("v_mov_b32 v76, 0" ::: "v76");
    lds[threadIdx.x] = threadIdx.x;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u *p = reinterpret_cast<const v4u *>(buf) + wave * n16_per_wave + (threadIdx.x & 63);
    unsigned acc = 0;
    for (int r = 0; r < reps; ++r)
        for (size_t i = 0; i + 8 * 64 <= n16_per_wave; i += 8 * 64) {
            v4u v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(p + i + k * 64);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
        }
    if (acc == 0x12345u) atomicAdd(sink, lds[(threadIdx.x + 1) & 255]);
}
static const uint4 *g_buf = nullptr;
static size_t g_n16_per_wave = 0;
static int g_stream = 0;
template <int V, bool SCRATCH>
__global__ void tiny(unsigned *sink, int n) {
    extern __shared__ unsigned lds[];
    if (V > 64) asm volatile("v_mov_b32 v161, 0" ::: "v161");
    if (SCRATCH) {
        volatile unsigned spill[40];
        for (int i = 0; i < n; ++i) spill[i] = i;
        if (n == 12345) atomicAdd(sink, spill[n & 31]);
    }
    if (n == 12345) { lds[threadIdx.x] = 1; atomicAdd(sink, lds[0]); }
}
#define CK(x) do { hipError_t s_ = (x); if (s_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(s_)); return 1; } } while (0)

static int g_delay_us = 60;
template <int V, bool SCRATCH>
int run(hipStream_t s0, hipStream_t s1, unsigned *sink, int ncu, size_t lds_a, size_t lds_b, int threads_b, int wg_b_per_cu) {
    hipEvent_t b0, b1, a0, a1;
    CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1)); CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
    CK(hipFuncSetAttribute((const void *)hog, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)stream_hog, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)tiny<V, SCRATCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a0, s0));
    if (g_stream) hipLaunchKernelGGL(stream_hog, dim3(ncu * 2), dim3(256), lds_a, s0, g_buf, g_n16_per_wave, 1, sink);
    else hipLaunchKernelGGL(hog, dim3(ncu * 2), dim3(256), lds_a, s0, 300ull * 100ull, sink);
    CK(hipEventRecord(a1, s0));
    if (g_delay_us) std::this_thread::sleep_for(std::chrono::microseconds(g_delay_us));
    CK(hipEventRecord(b0, s1));
    hipLaunchKernelGGL((tiny<V, SCRATCH>), dim3(ncu * wg_b_per_cu), dim3(threads_b), lds_b, s1, sink, 0);
    CK(hipEventRecord(b1, s1));
    CK(hipDeviceSynchronize());
    float tb = 0, ta = 0, off = 0;
    CK(hipEventElapsedTime(&tb, b0, b1)); CK(hipEventElapsedTime(&ta, a0, a1)); CK(hipEventElapsedTime(&off, a0, b0));
    printf("A lds %6zu | B: %3d thr, lds %6zu, vgpr>=%3d, scratch %d, %d wg/CU : B took %7.1f us (launched %.0f us after A; A took %.0f us)\n", lds_a, threads_b,
           lds_b, V, (int)SCRATCH, wg_b_per_cu, tb * 1e3, off * 1e3, ta * 1e3);
    return 0;
}

int main() {
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    unsigned *sink;
    CK(hipMalloc(&sink, 4));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("CUs %d, LDS per CU (maxSharedMemoryPerMultiProcessor) %zu\n", ncu, (size_t)prop.maxSharedMemoryPerMultiProcessor);
    run<162, true>(s0, s1, sink, ncu, 66048, 21504, 256, 1);  // first use of scratch on this queue
    for (int rep = 0; rep < 2; ++rep) {
        g_delay_us = rep ? 0 : 60;
        for (size_t lb : {0ul, 18944ul, 21504ul, 30720ul, 32768ul}) run<32, false>(s0, s1, sink, ncu, 66048, lb, 128, 1);
        run<162, true>(s0, s1, sink, ncu, 66048, 18944, 128, 1);
        run<162, true>(s0, s1, sink, ncu, 66048, 21504, 256, 1);
        run<162, true>(s0, s1, sink, ncu, 66048, 21504, 256, 3);
    }
    run<32, false>(s0, s1, sink, ncu, 66048, 21504, 256, 1);
    run<162, false>(s0, s1, sink, ncu, 66048, 21504, 256, 1);
    run<162, true>(s0, s1, sink, ncu, 66048, 21504, 256, 1);
    run<162, true>(s0, s1, sink, ncu, 66048, 21504, 256, 3);
    run<162, true>(s0, s1, sink, ncu, 49408, 21504, 256, 1);
    run<162, true>(s0, s1, sink, ncu, 49408, 39168, 256, 1);
    run<32, false>(s0, s1, sink, ncu, 66048, 0, 256, 4);
    // the same beside a kernel that streams HBM flat out (2 GB, ~300 us)
    {
        const size_t bytes = 2ull << 30;
        uint4 *buf;
        CK(hipMalloc(&buf, bytes));
        CK(hipMemset(buf, 1, bytes));
        g_buf = buf; g_n16_per_wave = bytes / 16 / ((size_t)ncu * 2 * 4); g_stream = 1;
        for (int rep = 0; rep < 2; ++rep) {
            g_delay_us = rep ? 60 : 0;
            run<32, false>(s0, s1, sink, ncu, 66048, 0, 128, 1);
            run<32, false>(s0, s1, sink, ncu, 66048, 18944, 128, 1);
            run<162, true>(s0, s1, sink, ncu, 66048, 18944, 128, 1);
            run<162, true>(s0, s1, sink, ncu, 66048, 21504, 256, 1);
            run<162, true>(s0, s1, sink, ncu, 66048, 21504, 256, 3);
        }
    }
    return 0;
}
