// LDS throughput micro-benchmark for gfx950: 256-thread workgroups, each thread writes 16 and reads 16 values per
// "exchange" (the Stockham exchange shape), unit-stride across lanes (conflict-free), as b32 / b64 / b128 accesses,
// with and without the two barriers, at 1..4 workgroups per CU.  Prints LDS bytes per nominal cycle per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int W, bool BAR>   // W = bytes per lane per access (8, 16)
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t base = (uint32_t)(uintptr_t)smem;
    const int tid = threadIdx.x;
    constexpr int NACC = 128 / W;                 // accesses per thread per direction: 128 B per thread per direction
    const uint32_t aw = base + tid * W;           // access s adds s*256*W
    const uint32_t ar = base + ((tid + 64) & 255) * W;   // read another wave's data
    float acc = 0.f;
    v4f val = {(float)tid, tid + 1.f, tid + 2.f, tid + 3.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < NACC; ++s) {
            if constexpr (W == 8) { v2f v = {val.x, val.y}; asm volatile("ds_write_b64 %0, %1" ::"v"(aw + s * 256 * W), "v"(v) : "memory"); }
            else asm volatile("ds_write_b128 %0, %1" ::"v"(aw + s * 256 * W), "v"(val) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (BAR) __syncthreads();
        if constexpr (W == 8) {
            v2f r[NACC];
#pragma unroll
            for (int s = 0; s < NACC; ++s) asm volatile("ds_read_b64 %0, %1" : "=v"(r[s]) : "v"(ar + s * 256 * W) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                         "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])::"memory");
#pragma unroll
            for (int s = 0; s < NACC; ++s) acc += r[s].x;
        } else {
            v4f r[NACC];
#pragma unroll
            for (int s = 0; s < NACC; ++s) asm volatile("ds_read_b128 %0, %1" : "=v"(r[s]) : "v"(ar + s * 256 * W) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])::"memory");
#pragma unroll
            for (int s = 0; s < NACC; ++s) acc += r[s].x;
        }
        val.x += acc;
        if (BAR) __syncthreads();
    }
    out[blockIdx.x * 256 + tid] = acc + val.x;
}

template <int W, bool BAR> void run(int wgs_per_cu, float *d) {
    const int iters = 2000, blocks = 256 * wgs_per_cu;
    const size_t lds = 160 * 1024 / wgs_per_cu > 65536 ? 65536 : 160 * 1024 / wgs_per_cu;   // pins residency (>= 32 KiB used)
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<W, BAR>), dim3(blocks), dim3(256), lds, 0, d, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<W, BAR>), dim3(blocks), dim3(256), lds, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes_per_cu = (double)iters * wgs_per_cu * 256.0 * 256.0;   // 128 B written + 128 B read per thread per iter
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("b%-3d %s  WG/CU=%d  %.3f ms  %.1f B/nominal-cycle/CU   (%.0f cycles per 64 KiB exchange)\n", W * 8, BAR ? "barriers" : "no-bar  ",
           wgs_per_cu, ms, bytes_per_cu / cyc, cyc / (iters * wgs_per_cu));
}

int main() {
    float *d;
    (void)hipMalloc(&d, sizeof(float) * 256 * 256 * 8);
    for (int w : {1, 2, 3, 4}) {
        run<8, false>(w, d);
        run<8, true>(w, d);
        run<16, false>(w, d);
        run<16, true>(w, d);
    }
    return 0;
}
