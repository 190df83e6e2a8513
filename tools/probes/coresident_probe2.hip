// coresident_probe2.hip -- the two-lane question of round 3: does a solve workgroup (256 threads, 21.5 KB of LDS, 168 VGPRs,
// scratch) get onto a CU that holds ONE streaming workgroup of 512 threads and 133 KB of LDS (bv_p1s_stream_kernel<8, 3, 2>)?
// Kernel A: 1 workgroup per CU, spins for 300 us.  20-60 us after its launch kernel B is launched on another stream and does
// nothing; its duration tells whether its workgroups had to wait for A's to leave.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>

__global__ __launch_bounds__(512) void hog512(unsigned long long ticks, unsigned *sink) {
    extern __shared__ unsigned lds[];
    asm volatile("v_mov_b32 v76, 0" ::: "v76");  // ~80 VGPRs, as the streaming kernel
    lds[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (lds[(threadIdx.x + 1) & 511] == 12345u) atomicAdd(sink, 1u);
}
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

template <int V, bool SCRATCH>
int run(hipStream_t s0, hipStream_t s1, unsigned *sink, int ncu, size_t lds_a, size_t lds_b, int threads_b, int wg_b_per_cu, int delay_us) {
    hipEvent_t b0, b1, a0, a1;
    CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1)); CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
    CK(hipFuncSetAttribute((const void *)hog512, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)tiny<V, SCRATCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a0, s0));
    hipLaunchKernelGGL(hog512, dim3(ncu), dim3(512), lds_a, s0, 300ull * 100ull, sink);
    CK(hipEventRecord(a1, s0));
    if (delay_us) std::this_thread::sleep_for(std::chrono::microseconds(delay_us));
    CK(hipEventRecord(b0, s1));
    hipLaunchKernelGGL((tiny<V, SCRATCH>), dim3(ncu * wg_b_per_cu), dim3(threads_b), lds_b, s1, sink, 0);
    CK(hipEventRecord(b1, s1));
    CK(hipDeviceSynchronize());
    float tb = 0, ta = 0, off = 0;
    CK(hipEventElapsedTime(&tb, b0, b1)); CK(hipEventElapsedTime(&ta, a0, a1)); CK(hipEventElapsedTime(&off, a0, b0));
    printf("A 512 thr, lds %6zu | B: %3d thr, lds %6zu, vgpr>=%3d, scratch %d, %d wg/CU : B took %7.1f us (launched %.0f us after A; A took %.0f us)\n", lds_a,
           threads_b, lds_b, V, (int)SCRATCH, wg_b_per_cu, tb * 1e3, off * 1e3, ta * 1e3);
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
    printf("CUs %d\n", ncu);
    run<162, true>(s0, s1, sink, ncu, 136192, 21504, 256, 1, 60);  // first use of scratch on this queue
    for (int rep = 0; rep < 2; ++rep) {
        for (size_t la : {136192ul, 131072ul, 122880ul, 98304ul}) {
            run<32, false>(s0, s1, sink, ncu, la, 0, 128, 1, 60);
            run<32, false>(s0, s1, sink, ncu, la, 18944, 128, 1, 60);
            run<162, true>(s0, s1, sink, ncu, la, 18944, 128, 1, 60);
            run<162, false>(s0, s1, sink, ncu, la, 21504, 256, 1, 60);
            run<162, true>(s0, s1, sink, ncu, la, 21504, 256, 1, 60);
            run<162, true>(s0, s1, sink, ncu, la, 21504, 256, 3, 60);
        }
    }
    return 0;
}
