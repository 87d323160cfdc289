// Can LDS traffic and VALU work overlap on one CU of gfx950?  Two workgroups per CU (256 threads each):
//   mode 0: both do only FMAs;  mode 1: both do only LDS exchanges (16 x ds_write_b64 + 16 x ds_read_b64 per thread);
//   mode 2: even blocks do FMAs, odd blocks LDS (different waves of the same SIMD);
//   mode 3: every wave alternates: one exchange, then the same number of FMAs as mode 0 per iteration.
// Work per block-iteration: NF FMAs per thread (wave-instructions) and/or one 64 KiB exchange.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float v2f __attribute__((ext_vector_type(2)));

#define FMA8 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"

template <int NF8> __device__ __forceinline__ void do_fma(float (&x)[8], float a, float b) {
#pragma unroll
    for (int u = 0; u < NF8; ++u)
        asm volatile(FMA8 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b));
}
template <bool BAR = true> __device__ __forceinline__ void do_lds(uint32_t aw, uint32_t ar, float (&x)[8]) {
    v2f val = {x[0], x[1]};
#pragma unroll
    for (int s = 0; s < 16; ++s) asm volatile("ds_write_b64 %0, %1" ::"v"(aw + s * 2048), "v"(val) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (BAR) __syncthreads();
    v2f r[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) asm volatile("ds_read_b64 %0, %1" : "=v"(r[s]) : "v"(ar + s * 2048) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                 "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])::"memory");
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc += r[s].x;
    x[0] += acc * 1e-30f;
    if (BAR) __syncthreads();
}

template <int MODE, int NF8> __global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t base = (uint32_t)(uintptr_t)smem;
    const int tid = threadIdx.x;
    const uint32_t aw = base + tid * 8, ar = base + ((tid + 64) & 255) * 8;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = tid + i;
    const bool odd = blockIdx.x & 1;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) do_fma<NF8>(x, a, b);
        else if (MODE == 1) do_lds(aw, ar, x);
        else if (MODE == 2) {
            if (odd) do_lds(aw, ar, x);
            else do_fma<NF8>(x, a, b);
        } else {
            do_lds(aw, ar, x);
            do_fma<NF8>(x, a, b);
        }
    }
    out[blockIdx.x * 256 + tid] = x[0] + x[1] + x[2] + x[3] + x[4] + x[5] + x[6] + x[7];
}

// mode 4/5/6: 512-thread workgroups, one per CU-slot; waves 0-3 (one per SIMD) and waves 4-7 (one per SIMD) take roles:
//   4: FMA | FMA    5: LDS | LDS (no barriers)    6: FMA | LDS
template <int MODE, int NF8> __global__ __launch_bounds__(512) void k8(float *out, int iters, float a, float b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t base = (uint32_t)(uintptr_t)smem;
    const int tid = threadIdx.x;
    const uint32_t aw = base + (tid & 255) * 8 + (tid >> 8) * 32768, ar = base + ((tid + 64) & 255) * 8 + (tid >> 8) * 32768;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = tid + i;
    const bool hi = tid >= 256;
    const bool lds_role = MODE == 5 || (MODE == 6 && hi);
    for (int i = 0; i < iters; ++i) {
        if (lds_role) do_lds<false>(aw, ar, x);
        else do_fma<NF8>(x, a, b);
    }
    out[blockIdx.x * 512 + tid] = x[0] + x[1] + x[2] + x[3] + x[4] + x[5] + x[6] + x[7];
}
template <int MODE, int NF8> float run8(float *d, int wgs_per_cu) {
    const int iters = 2000, blocks = 256 * wgs_per_cu;
    const size_t lds = 65536;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k8<MODE, NF8>), dim3(blocks), dim3(512), lds, 0, d, 10, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k8<MODE, NF8>), dim3(blocks), dim3(512), lds, 0, d, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int MODE, int NF8> float run(float *d, int wgs_per_cu) {
    const int iters = 2000, blocks = 256 * wgs_per_cu;
    const size_t lds = 160 * 1024 / wgs_per_cu > 65536 ? 65536 : 160 * 1024 / wgs_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NF8>), dim3(blocks), dim3(256), lds, 0, d, 10, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NF8>), dim3(blocks), dim3(256), lds, 0, d, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float *d;
    (void)hipMalloc(&d, sizeof(float) * 256 * 256 * 8);
    constexpr int NF8 = 32;   // 256 FMAs per thread per iteration ~ one radix-16 pass + twiddles
    for (int w : {2, 4}) {
        const float t0 = run<0, NF8>(d, w), t1 = run<1, NF8>(d, w), t2 = run<2, NF8>(d, w), t3 = run<3, NF8>(d, w);
        printf("WG/CU=%d  (2000 iterations; 256 FMAs per thread and/or one 64 KiB exchange per workgroup-iteration)\n", w);
        printf("  all FMA            %.3f ms\n  all LDS            %.3f ms\n", t0, t1);
        printf("  half FMA, half LDS %.3f ms   (perfect overlap: %.3f, serial: %.3f)\n", t2, (t0 > t1 ? t0 : t1) / 2, (t0 + t1) / 2);
        printf("  every wave LDS+FMA %.3f ms   (perfect overlap: %.3f, serial: %.3f)\n", t3, t0 > t1 ? t0 : t1, t0 + t1);
    }
    for (int w : {1, 2}) {
        const float t4 = run8<4, NF8>(d, w), t5 = run8<5, NF8>(d, w), t6 = run8<6, NF8>(d, w);
        printf("512-thread WG x %d per CU: waves 0-3 | waves 4-7 (one of each per SIMD), no barriers\n", w);
        printf("  FMA|FMA %.3f ms   LDS|LDS %.3f ms   FMA|LDS %.3f ms  (overlap => max(FMA|FMA at half load, LDS|LDS at half load))\n", t4, t5, t6);
    }
    return 0;
}
