// stream_mlp.hip -- how much of the HBM streaming rate a CU reaches as a function of the bytes it keeps in flight.
// One workgroup of 256 threads per slot reads its own contiguous region in 16 KiB chunks (thread t takes the 8-byte
// words t + 256 s of a chunk, s = 0..7 -- the access pattern of the Welch front role), DEPTH chunks ahead of the one it
// sums.  Grid = 256 * WGS workgroups (WGS resident per CU).  Prints TB/s per (WGS, DEPTH).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/stream_mlp.hip -o tools/ubench/stream_mlp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int DEPTH>
__global__ __launch_bounds__(256) void k_stream(const float2 *__restrict__ x, long chunks_per_wg, float *__restrict__ out) {
    const float2 *base = x + (long)blockIdx.x * chunks_per_wg * 2048;
    float2 buf[DEPTH][8];
    float ax = 0.f, ay = 0.f;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int s = 0; s < 8; ++s) buf[d][s] = base[(long)d * 2048 + threadIdx.x + 256 * s];
    for (long c = 0; c < chunks_per_wg; c += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                ax += buf[d][s].x;
                ay += buf[d][s].y;
            }
            long nc = c + d + DEPTH;
            nc = nc < chunks_per_wg ? nc : chunks_per_wg - 1;
#pragma unroll
            for (int s = 0; s < 8; ++s) buf[d][s] = base[nc * 2048 + threadIdx.x + 256 * s];
        }
    }
    if (ax + ay == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = ax;
}

template <int DEPTH> static float run(const float2 *x, long total_chunks, int wgs, float *out, size_t lds_pad) {
    const int grid = 256 * wgs;
    const long cpw = total_chunks / grid;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k_stream<DEPTH>, dim3(grid), dim3(256), lds_pad, 0, x, cpw, out);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        if (r >= 2 && ms < best) best = ms;
    }
    return best;
}

int main() {
    const long n = 1L << 28;                       // complex64 samples = 2 GiB
    float2 *x;
    float *out;
    (void)hipMalloc(&x, n * sizeof(float2));
    (void)hipMalloc(&out, 1 << 22);
    std::vector<float> h(1 << 20);
    for (auto &v : h) v = (float)(rand() & 0xffff) / 65536.f;
    for (long o = 0; o < n * 2; o += (1 << 20)) (void)hipMemcpy((float *)x + o, h.data(), sizeof(float) << 20, hipMemcpyHostToDevice);
    const long chunks = n / 2048;
    printf("2 GiB read once; rows: workgroups of 256 threads resident per CU (LDS-padded), cols: 16 KiB chunks in flight per workgroup\n");
    printf("%-8s %10s %10s %10s %10s\n", "wgs/CU", "depth1", "depth2", "depth4", "depth6");
    for (int wgs : {1, 2, 3, 4, 8}) {
        const size_t pad = wgs >= 8 ? 16 * 1024 : (size_t)(150 * 1024 / wgs);       // pins the residency
        (void)hipFuncSetAttribute((const void *)k_stream<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)k_stream<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)k_stream<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)k_stream<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        const float t1 = run<1>(x, chunks, wgs, out, pad), t2 = run<2>(x, chunks, wgs, out, pad), t4 = run<4>(x, chunks, wgs, out, pad),
                    t6 = run<6>(x, chunks, wgs, out, pad);
        const double gb = n * 8.0 / 1e9;
        printf("%-8d %7.2f TB/s %7.2f TB/s %7.2f TB/s %7.2f TB/s   (%.3f %.3f %.3f %.3f ms)\n", wgs, gb / t1, gb / t2, gb / t4, gb / t6, t1, t2, t4, t6);
    }
    return 0;
}
