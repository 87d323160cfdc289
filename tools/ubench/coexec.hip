// coexec.hip -- do two kernels from two HIP streams share the CUs of gfx950 when the first one leaves room?
// P mimics k_welch_pipe's footprint: 256 workgroups of 768 threads, 128.5 KiB of dynamic LDS, ~130 VGPRs, busy for ~T_P us.
// Q mimics a light epilogue: 256 workgroups of 256 threads, VQ VGPRs (launch-bounds controlled), 4 KiB LDS, busy ~T_Q us.
// Timed (HIP events on a third stream / host clock): P alone, Q alone, P then Q on one stream, P on stream A with Q on stream B
// (Q launched right after P, and Q launched first).  If the pair takes ~max(P, Q) the dispatcher co-schedules them.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/coexec.hip -o tools/ubench/coexec
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CHK(e)                                                                                        \
    do {                                                                                              \
        hipError_t r_ = (e);                                                                          \
        if (r_ != hipSuccess) {                                                                       \
            printf("%s failed: %s\n", #e, hipGetErrorString(r_));                                     \
            exit(1);                                                                                  \
        }                                                                                             \
    } while (0)

template <int NREG>
__device__ __forceinline__ float spin(float seed, long iters) {
    float r[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) r[i] = seed + i;
    for (long it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NREG; ++i) r[i] = fmaf(r[i], 1.0000001f, 0.5f);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NREG; ++i) s += r[i];
    return s;
}

__global__ __launch_bounds__(768) void k_big(float *out, long iters) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const float s = spin<120>(lds[(threadIdx.x + 1) % 768], iters);
    if (s == 12345.f) out[blockIdx.x] = s;
}
template <int NREG, int BOUND>
__global__ __launch_bounds__(256, BOUND) void k_small(float *out, long iters) {
    __shared__ float lds[1024];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const float s = spin<NREG>(lds[(threadIdx.x + 1) % 256], iters);
    if (s == 12345.f) out[blockIdx.x] = s;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    float *out;
    CHK(hipMalloc(&out, 4096));
    hipStream_t a, b;
    CHK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    const size_t lds = 131584;
    CHK(hipFuncSetAttribute((const void *)k_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long itP = 600, itQ = 400;
    auto P = [&](hipStream_t s) { hipLaunchKernelGGL(k_big, dim3(256), dim3(768), lds, s, out, itP); };
    auto Q48 = [&](hipStream_t s) { hipLaunchKernelGGL((k_small<40, 4>), dim3(256), dim3(256), 0, s, out, itQ * 3); };
    auto Q160 = [&](hipStream_t s) { hipLaunchKernelGGL((k_small<150, 2>), dim3(256), dim3(256), 0, s, out, itQ); };
    auto timeit = [&](const char *name, auto fn) {
        for (int w = 0; w < 3; ++w) fn();
        CHK(hipDeviceSynchronize());
        const int reps = 20;
        const double t0 = now_us();
        for (int r = 0; r < reps; ++r) fn();
        CHK(hipDeviceSynchronize());
        printf("%-46s %8.1f us per round\n", name, (now_us() - t0) / reps);
    };
    for (int round = 0; round < 2; ++round) {
        timeit("P alone", [&] { P(a); });
        timeit("Q (<= 128 VGPR, 40 live) alone", [&] { Q48(a); });
        timeit("Q (~160 VGPR) alone", [&] { Q160(a); });
        timeit("P, Q-light on ONE stream", [&] { P(a); Q48(a); });
        timeit("P on A, Q-light on B (P first)", [&] { P(a); Q48(b); });
        timeit("Q-light on B, P on A (Q first)", [&] { Q48(b); P(a); });
        timeit("P, Q-heavy on ONE stream", [&] { P(a); Q160(a); });
        timeit("P on A, Q-heavy on B (P first)", [&] { P(a); Q160(b); });
        // the pipelined pattern: B's Q waits for the PREVIOUS P (event), A runs P back to back
        hipEvent_t ev;
        CHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        timeit("A: P,P,P,P  B: (wait P_k) Q-light, x4  [per 4]", [&] {
            for (int k = 0; k < 4; ++k) {
                P(a);
                CHK(hipEventRecord(ev, a));
                CHK(hipStreamWaitEvent(b, ev, 0));
                Q48(b);
            }
        });
        timeit("one stream: P,Q-light x4  [per 4]", [&] {
            for (int k = 0; k < 4; ++k) {
                P(a);
                Q48(a);
            }
        });
        CHK(hipEventDestroy(ev));
    }
    return 0;
}
