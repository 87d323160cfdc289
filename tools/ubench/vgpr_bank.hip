// Does the VGPR bank of the source operands change the VALU issue rate on gfx950?  Independent v_fma_f32 / v_add_f32
// with hand-picked source registers (bank = register number mod 4), 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(X) X X X X X X X X
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, int iters) {
    // sources v0..v11 are set once; destinations v16..v23 (never read)
    asm volatile("v_mov_b32 v0, 1.0\n v_mov_b32 v1, 1.0\n v_mov_b32 v2, 1.0\n v_mov_b32 v3, 1.0\n v_mov_b32 v4, 1.0\n v_mov_b32 v5, 1.0\n"
                 "v_mov_b32 v6, 1.0\n v_mov_b32 v7, 1.0\n v_mov_b32 v8, 1.0\n v_mov_b32 v9, 1.0\n v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n" ::
                     : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11");
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0)   // three different banks
            asm volatile(REP8("v_fma_f32 v16, v0, v1, v2\n v_fma_f32 v17, v1, v2, v3\n v_fma_f32 v18, v2, v3, v4\n v_fma_f32 v19, v3, v4, v5\n"
                              "v_fma_f32 v20, v4, v5, v6\n v_fma_f32 v21, v5, v6, v7\n v_fma_f32 v22, v6, v7, v8\n v_fma_f32 v23, v7, v8, v9\n")
                         ::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23");
        else if (MODE == 1)   // all three in one bank
            asm volatile(REP8("v_fma_f32 v16, v0, v4, v8\n v_fma_f32 v17, v1, v5, v9\n v_fma_f32 v18, v2, v6, v10\n v_fma_f32 v19, v3, v7, v11\n"
                              "v_fma_f32 v20, v4, v8, v0\n v_fma_f32 v21, v5, v9, v1\n v_fma_f32 v22, v6, v10, v2\n v_fma_f32 v23, v7, v11, v3\n")
                         ::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23");
        else if (MODE == 2)   // two in one bank
            asm volatile(REP8("v_fma_f32 v16, v0, v4, v1\n v_fma_f32 v17, v1, v5, v2\n v_fma_f32 v18, v2, v6, v3\n v_fma_f32 v19, v3, v7, v0\n"
                              "v_fma_f32 v20, v4, v8, v5\n v_fma_f32 v21, v5, v9, v6\n v_fma_f32 v22, v6, v10, v7\n v_fma_f32 v23, v7, v11, v4\n")
                         ::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23");
        else if (MODE == 3)   // add, different banks
            asm volatile(REP8("v_add_f32 v16, v0, v1\n v_add_f32 v17, v1, v2\n v_add_f32 v18, v2, v3\n v_add_f32 v19, v3, v4\n"
                              "v_add_f32 v20, v4, v5\n v_add_f32 v21, v5, v6\n v_add_f32 v22, v6, v7\n v_add_f32 v23, v7, v8\n")
                         ::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23");
        else if (MODE == 4)   // add, same bank
            asm volatile(REP8("v_add_f32 v16, v0, v4\n v_add_f32 v17, v1, v5\n v_add_f32 v18, v2, v6\n v_add_f32 v19, v3, v7\n"
                              "v_add_f32 v20, v4, v8\n v_add_f32 v21, v5, v9\n v_add_f32 v22, v6, v10\n v_add_f32 v23, v7, v11\n")
                         ::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23");
        else if (MODE == 5)   // fma with an SGPR multiplier (constant bus) + two VGPRs in different banks
            asm volatile(REP8("v_fma_f32 v16, s20, v1, v2\n v_fma_f32 v17, s20, v2, v3\n v_fma_f32 v18, s20, v3, v4\n v_fma_f32 v19, s20, v4, v5\n"
                              "v_fma_f32 v20, s20, v5, v6\n v_fma_f32 v21, s20, v6, v7\n v_fma_f32 v22, s20, v7, v8\n v_fma_f32 v23, s20, v8, v9\n")
                         ::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "s20");
        else if (MODE == 7)   // VOP2 mul with a 32-bit literal
            asm volatile(REP8("v_mul_f32 v16, 0x3f3504f3, v1\n v_mul_f32 v17, 0x3f3504f3, v2\n v_mul_f32 v18, 0x3f3504f3, v3\n v_mul_f32 v19, 0x3f3504f3, v4\n"
                              "v_mul_f32 v20, 0x3f3504f3, v5\n v_mul_f32 v21, 0x3f3504f3, v6\n v_mul_f32 v22, 0x3f3504f3, v7\n v_mul_f32 v23, 0x3f3504f3, v8\n")
                         ::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23");
        else if (MODE == 8)   // VOP2 add with an SGPR
            asm volatile(REP8("v_add_f32 v16, s20, v1\n v_add_f32 v17, s20, v2\n v_add_f32 v18, s20, v3\n v_add_f32 v19, s20, v4\n"
                              "v_add_f32 v20, s20, v5\n v_add_f32 v21, s20, v6\n v_add_f32 v22, s20, v7\n v_add_f32 v23, s20, v8\n")
                         ::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "s20");
        else if (MODE == 9)   // fma with an inline constant
            asm volatile(REP8("v_fma_f32 v16, 2.0, v1, v2\n v_fma_f32 v17, 2.0, v2, v3\n v_fma_f32 v18, 2.0, v3, v4\n v_fma_f32 v19, 2.0, v4, v5\n"
                              "v_fma_f32 v20, 2.0, v5, v6\n v_fma_f32 v21, 2.0, v6, v7\n v_fma_f32 v22, 2.0, v7, v8\n v_fma_f32 v23, 2.0, v8, v9\n")
                         ::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23");
        else if (MODE == 10)   // v_fmamk_f32: d = s0 * K + s1
            asm volatile(REP8("v_fmamk_f32 v16, v1, 0x3f3504f3, v2\n v_fmamk_f32 v17, v2, 0x3f3504f3, v3\n v_fmamk_f32 v18, v3, 0x3f3504f3, v4\n v_fmamk_f32 v19, v4, 0x3f3504f3, v5\n"
                              "v_fmamk_f32 v20, v5, 0x3f3504f3, v6\n v_fmamk_f32 v21, v6, 0x3f3504f3, v7\n v_fmamk_f32 v22, v7, 0x3f3504f3, v8\n v_fmamk_f32 v23, v8, 0x3f3504f3, v9\n")
                         ::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23");
        else if (MODE == 11)   // fma with source modifiers (neg) on VGPRs
            asm volatile(REP8("v_fma_f32 v16, -v0, v1, v2\n v_fma_f32 v17, -v1, v2, v3\n v_fma_f32 v18, -v2, v3, v4\n v_fma_f32 v19, -v3, v4, v5\n"
                              "v_fma_f32 v20, -v4, v5, v6\n v_fma_f32 v21, -v5, v6, v7\n v_fma_f32 v22, -v6, v7, v8\n v_fma_f32 v23, -v7, v8, v9\n")
                         ::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23");
        else if (MODE == 12)   // VOP2 mul with an inline constant
            asm volatile(REP8("v_mul_f32 v16, 0.5, v1\n v_mul_f32 v17, 0.5, v2\n v_mul_f32 v18, 0.5, v3\n v_mul_f32 v19, 0.5, v4\n"
                              "v_mul_f32 v20, 0.5, v5\n v_mul_f32 v21, 0.5, v6\n v_mul_f32 v22, 0.5, v7\n v_mul_f32 v23, 0.5, v8\n")
                         ::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23");
        else   // v_fmac (VOP2: dst is the addend), sources in different banks
            asm volatile(REP8("v_fmac_f32 v16, v0, v1\n v_fmac_f32 v17, v1, v2\n v_fmac_f32 v18, v2, v3\n v_fmac_f32 v19, v3, v4\n"
                              "v_fmac_f32 v20, v4, v5\n v_fmac_f32 v21, v5, v6\n v_fmac_f32 v22, v6, v7\n v_fmac_f32 v23, v7, v8\n")
                         ::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23");
    }
    float r;
    asm volatile("v_add_f32 %0, v16, v17" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> void run(const char *name, float *d, int wps) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, d, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 64.0;
    printf("%-34s waves/SIMD=%d  %.3f ms  %.2f nominal cycles per instr per SIMD\n", name, wps, ms, ms * 1e-3 * 2.4e9 / (n * wps));
}

int main() {
    float *d;
    (void)hipMalloc(&d, sizeof(float) * 256 * 256 * 8);
    for (int wps : {2, 4}) {
        run<0>("fma  3 banks", d, wps);
        run<1>("fma  1 bank", d, wps);
        run<2>("fma  2 in one bank", d, wps);
        run<3>("add  2 banks", d, wps);
        run<4>("add  1 bank", d, wps);
        run<5>("fma  sgpr x vgpr + vgpr", d, wps);
        run<6>("fmac 2 banks (+dst)", d, wps);
        run<7>("mul  literal x vgpr", d, wps);
        run<8>("add  sgpr + vgpr", d, wps);
        run<9>("fma  inline-const x vgpr + vgpr", d, wps);
        run<10>("fmamk vgpr x literal + vgpr", d, wps);
        run<11>("fma  -vgpr x vgpr + vgpr", d, wps);
        run<12>("mul  inline-const x vgpr", d, wps);
    }
    return 0;
}
