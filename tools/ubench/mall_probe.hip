// mall_probe.hip -- can one kernel hand a buffer to the next through the memory-side cache (256 MiB "Infinity Cache") of gfx950?
// For buffer sizes 16 MiB .. 1 GiB: the rate of a streaming READ kernel over the buffer
//   cold          after a 2 GiB sweep of another buffer (nothing of it cached anywhere),
//   after a read  of the same buffer (read allocation),
//   after a write of the same buffer by the preceding kernel (write allocation -- the hand-over cfg5 / the long transforms would need),
// and the rate of the write kernel itself.  All launches on one stream, HIP events around the measured kernel only.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mall_probe.hip -o tools/ubench/mall_probe
#include <hip/hip_runtime.h>
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

__global__ __launch_bounds__(256) void k_read(const float4 *__restrict__ p, size_t n16, float *sink) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const float4 q0 = p[i], q1 = p[i + stride], q2 = p[i + 2 * stride], q3 = p[i + 3 * stride];
        a.x += q0.x + q1.x + q2.x + q3.x;
        a.y += q0.y + q1.y + q2.y + q3.y;
        a.z += q0.z + q1.z + q2.z + q3.z;
        a.w += q0.w + q1.w + q2.w + q3.w;
    }
    for (; i < n16; i += stride) {
        const float4 q = p[i];
        a.x += q.x;
        a.y += q.y;
        a.z += q.z;
        a.w += q.w;
    }
    if (a.x + a.y + a.z + a.w == 123.456f) sink[0] = 1.f;      // never true: keeps the loads
}

__global__ __launch_bounds__(256) void k_write(float4 *__restrict__ p, size_t n16, float v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
}

int main() {
    hipStream_t st;
    CHK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const size_t big = (size_t)2 << 30;
    float4 *flush, *buf;
    float *sink;
    CHK(hipMalloc(&flush, big));
    CHK(hipMalloc(&buf, (size_t)1 << 30));
    CHK(hipMalloc(&sink, 4));
    CHK(hipMemset(flush, 0, big));
    CHK(hipMemset(buf, 0, (size_t)1 << 30));
    const int grid = 256 * 8;
    auto timed = [&](auto &&launch) {
        CHK(hipEventRecord(e0, st));
        launch();
        CHK(hipEventRecord(e1, st));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        return ms;
    };
    auto sweep = [&]() { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, st, flush, big / 16, sink); };
    printf("%10s %12s %12s %12s %12s   (GB/s)\n", "MiB", "write", "read cold", "read>read", "write>read");
    for (size_t mib : {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024}) {
        const size_t bytes = mib << 20, n16 = bytes / 16;
        double best[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 4; ++rep) {
            sweep();
            const float tw = timed([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, st, buf, n16, (float)rep); });
            sweep();
            const float tc = timed([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, st, buf, n16, sink); });
            const float trr = timed([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, st, buf, n16, sink); });
            sweep();
            hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, st, buf, n16, (float)rep + 0.5f);
            const float twr = timed([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, st, buf, n16, sink); });
            const double g[4] = {bytes / tw / 1e6, bytes / tc / 1e6, bytes / trr / 1e6, bytes / twr / 1e6};
            for (int k = 0; k < 4; ++k)
                if (g[k] > best[k]) best[k] = g[k];
        }
        printf("%10zu %12.0f %12.0f %12.0f %12.0f\n", mib, best[0], best[1], best[2], best[3]);
    }
    return 0;
}
