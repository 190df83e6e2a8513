// xstream_probe.hip -- what a cross-stream dependency costs on this stack, and whether kernels of ONE stream can overlap
// (hipExtAnyOrderLaunch).  hipcc --offload-arch=gfx950 -O2 xstream_probe.hip -o xstream_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin_kernel(unsigned long long cycles, unsigned *sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(sink, 1u);
}
#define CK(x) do { hipError_t s_ = (x); if (s_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(s_)); return 1; } } while (0)

int main() {
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    unsigned *sink;
    CK(hipMalloc(&sink, 4));
    const unsigned long long us100 = 100ull * 100ull;  // wall_clock64 ticks at 100 MHz
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const int R = 200;
    for (unsigned flags : {0u, (unsigned)hipEventDisableTiming, (unsigned)(hipEventDisableTiming | hipEventReleaseToDevice)}) {
        std::vector<hipEvent_t> ev(4 * R);
        for (auto &e : ev) CK(hipEventCreateWithFlags(&e, flags));
        // (a) one stream: 2R kernels of 100 us back to back
        CK(hipDeviceSynchronize());
        auto t0 = now();
        for (int i = 0; i < 2 * R; ++i) hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, s0, us100, sink);
        CK(hipStreamSynchronize(s0));
        const double one = ms(t0, now());
        // (b) ping-pong: kernel on s0, event, s1 waits, kernel on s1, event, s0 waits, ...
        t0 = now();
        for (int i = 0; i < R; ++i) {
            hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, s0, us100, sink);
            CK(hipEventRecord(ev[2 * i], s0));
            CK(hipStreamWaitEvent(s1, ev[2 * i], 0));
            hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, s1, us100, sink);
            CK(hipEventRecord(ev[2 * i + 1], s1));
            CK(hipStreamWaitEvent(s0, ev[2 * i + 1], 0));
        }
        CK(hipStreamSynchronize(s0));
        const double pp = ms(t0, now());
        // (c) fork: kernel on s0; s1 waits and runs a kernel CONCURRENT with the next kernel of s0 (ideal: R x 100 us)
        t0 = now();
        for (int i = 0; i < R; ++i) {
            hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, s0, us100, sink);
            CK(hipEventRecord(ev[2 * i], s0));
            CK(hipStreamWaitEvent(s1, ev[2 * i], 0));
            hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, s1, us100 * 8 / 10, sink);
        }
        CK(hipStreamSynchronize(s0));
        CK(hipStreamSynchronize(s1));
        const double fork = ms(t0, now());
        printf("event flags %#x: %d kernels of 100 us on one stream %.3f ms | ping-pong over two streams %.3f ms (+%.1f us per hop) | fork %d x (100 us || 80 us) %.3f ms\n",
               flags, 2 * R, one, pp, (pp - one) / (2 * R) * 1e3, R, fork);
        for (auto &e : ev) CK(hipEventDestroy(e));
    }
    // (d) one stream, second kernel of each pair launched with hipExtAnyOrderLaunch: pairs overlap if the flag is honoured
    {
        CK(hipDeviceSynchronize());
        auto t0 = now();
        for (int i = 0; i < R; ++i) {
            hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, s0, us100, sink);
            hipExtLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, s0, nullptr, nullptr, hipExtAnyOrderLaunch, us100, sink);
        }
        CK(hipStreamSynchronize(s0));
        printf("one stream, every second kernel hipExtAnyOrderLaunch: %d kernels of 100 us %.3f ms (overlap => ~%d x 100 us)\n", 2 * R, ms(t0, now()), R);
    }
    return 0;
}
