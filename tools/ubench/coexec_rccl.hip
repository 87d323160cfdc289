// coexec_rccl.hip -- does leaving CUs free keep a foreign kernel that CANNOT share a CU with the main kernel from delaying it?
// P mimics k_welch_pipe on a 2^25-sample shard: G workgroups of 768 threads, 128.5 KiB dynamic LDS, ~130 VGPRs, ~70 us each, launched
// back to back on stream A.  R mimics RCCL's collective kernel: 4 workgroups of 256 threads with ~250 live VGPRs (one wave per SIMD,
// nothing fits beside a P workgroup), ~30 us, on stream B behind an event of the previous P -- the streaming engine's pattern.
// Compared, per P: P alone at G = 256; P + R at G = 256 (R must wait for / displace P workgroups); P + R at G = 248 with
// 256/248 of the work per workgroup (8 CUs left free).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/coexec_rccl.hip -o tools/ubench/coexec_rccl
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
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
__global__ __launch_bounds__(256, 1) void k_fat(float *out, long iters) {       // ~250 VGPRs: one wave per SIMD
    __shared__ float lds[4096];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const float s = spin<240>(lds[(threadIdx.x + 1) % 256], iters);
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
    hipEvent_t ev;
    CHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    // calibrate: iterations for ~70 us (P) and ~30 us (R)
    auto timeK = [&](auto launch) {
        for (int w = 0; w < 3; ++w) launch();
        CHK(hipDeviceSynchronize());
        const double t0 = now_us();
        for (int r = 0; r < 50; ++r) launch();
        CHK(hipDeviceSynchronize());
        return (now_us() - t0) / 50;
    };
    long itP = 200, itR = 100;
    for (int k = 0; k < 3; ++k) {
        const double tp = timeK([&] { hipLaunchKernelGGL(k_big, dim3(256), dim3(768), lds, a, out, itP); });
        itP = (long)(itP * 70.0 / tp);
        const double tr = timeK([&] { hipLaunchKernelGGL(k_fat, dim3(4), dim3(256), 0, a, out, itR); });
        itR = (long)(itR * 30.0 / tr);
    }
    printf("P alone (256 WG): %.1f us   R alone (4 WG): %.1f us\n",
           timeK([&] { hipLaunchKernelGGL(k_big, dim3(256), dim3(768), lds, a, out, itP); }),
           timeK([&] { hipLaunchKernelGGL(k_fat, dim3(4), dim3(256), 0, a, out, itR); }));
    // withR: 0 = P only; -1 = P + the event record / wait of the engine (no foreign kernel); n > 0 = R with n workgroups behind every P
    auto series = [&](const char *name, int G, long it, int withR) {
        auto run = [&](int n) {
            for (int k = 0; k < n; ++k) {
                hipLaunchKernelGGL(k_big, dim3(G), dim3(768), lds, a, out, it);
                if (withR) {
                    CHK(hipEventRecord(ev, a));
                    CHK(hipStreamWaitEvent(b, ev, 0));
                    if (withR > 0) hipLaunchKernelGGL(k_fat, dim3(withR), dim3(256), 0, b, out, itR);
                }
            }
        };
        run(20);
        CHK(hipDeviceSynchronize());
        const double t0 = now_us();
        run(200);
        CHK(hipDeviceSynchronize());
        printf("%-58s %8.1f us per step\n", name, (now_us() - t0) / 200);
    };
    // the event as the main kernel's own completion signal (hipExtLaunchKernelGGL stop event) instead of a record behind it
    auto series_ext = [&](const char *name, int G, long it, int withR) {
        auto run = [&](int n) {
            for (int k = 0; k < n; ++k) {
                hipExtLaunchKernelGGL(k_big, dim3(G), dim3(768), lds, a, nullptr, ev, 0, out, it);
                CHK(hipStreamWaitEvent(b, ev, 0));
                if (withR > 0) hipLaunchKernelGGL(k_fat, dim3(withR), dim3(256), 0, b, out, itR);
            }
        };
        run(20);
        CHK(hipDeviceSynchronize());
        const double t0 = now_us();
        run(200);
        CHK(hipDeviceSynchronize());
        printf("%-58s %8.1f us per step\n", name, (now_us() - t0) / 200);
    };
    for (int round = 0; round < 2; ++round) {
        series_ext("P x 256, stop event of the launch + wait only", 256, itP, -1);
        series_ext("P x 248, stop event of the launch, R (4 workgroups)", 248, itP * 256 / 248, 4);
        series("P x 256 workgroups, no R", 256, itP, 0);
        series("P x 256, event record + wait only", 256, itP, -1);
        series("P x 256, R (4 workgroups) behind every P", 256, itP, 4);
        series("P x 256, R (8 workgroups) behind every P", 256, itP, 8);
        series("P x 248 (x 256/248 work), no R", 248, itP * 256 / 248, 0);
        series("P x 248, event record + wait only", 248, itP * 256 / 248, -1);
        series("P x 248, R (4 workgroups)", 248, itP * 256 / 248, 4);
        series("P x 248, R (8 workgroups)", 248, itP * 256 / 248, 8);
        series("P x 248, R (16 workgroups)", 248, itP * 256 / 248, 16);
        series("P x 252 (x 256/252 work), R (4 workgroups)", 252, itP * 256 / 252, 4);
        series("P x 240 (x 256/240 work), R (16 workgroups)", 240, itP * 256 / 240, 16);
    }
    return 0;
}
