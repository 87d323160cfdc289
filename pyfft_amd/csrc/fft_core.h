// fft_core.h -- workgroup-level Stockham autosort FFT for gfx950 (wave64, LDS exchange).
//
// One "group" of T = N/R threads owns one N-point transform; every thread keeps R complex
// points in registers (R = 16 for N >= 16).  The data contract on entry AND exit is
//        v[t]  <->  element  (tid + T*t),   t = 0..R-1
// so global loads/stores are coalesced across the group for every t, a forward transform can
// be chained straight into an inverse one (Hilbert, FIR) with no reshuffle, and an overlapped
// frame can be carried from one Welch frame to the next inside the same registers.
//
// Passes: radix-16 as many times as they fit, then one radix-{2,4,8} remainder.  Between
// passes the group exchanges through LDS (Stockham: scattered write, unit-stride read):
//   - the first exchange (stride-16 scatter) uses a [s][q] image with row pitch T+1 complex
//     so that both the 16-lane ds_write_b64 groups and the 16-lane ds_read2_b64 groups (what
//     hipcc emits for the unit-stride reads) are bank-conflict free (bank = dword mod 32);
//   - later exchanges are conflict free in the plain linear image.
// Twiddles W_N^m come from a per-N global table (float, rounded from double on the host) and
// are held in registers across frames.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

namespace sp {

typedef float2 cf;

__device__ __forceinline__ cf mk(float a, float b) { return make_float2(a, b); }
__device__ __forceinline__ cf operator+(cf a, cf b) { return mk(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf operator-(cf a, cf b) { return mk(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cf operator*(float s, cf a) { return mk(s * a.x, s * a.y); }
__device__ __forceinline__ cf cmul(cf a, cf b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cf cmulc(cf a, cf b) { return mk(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a*conj(b)
__device__ __forceinline__ cf cconj(cf a) { return mk(a.x, -a.y); }
__device__ __forceinline__ float cnorm(cf a) { return a.x * a.x + a.y * a.y; }

// multiply by -i (forward) / +i (inverse)
template <bool INV> __device__ __forceinline__ cf rot(cf a) { return INV ? mk(-a.y, a.x) : mk(a.y, -a.x); }
// multiply by the compile-time constant (C, -S) forward, (C, +S) inverse, i.e. W = exp(-/+ i*theta)
template <bool INV> __device__ __forceinline__ cf twc(cf a, float C, float S) {
    return INV ? mk(a.x * C - a.y * S, a.y * C + a.x * S) : mk(a.x * C + a.y * S, a.y * C - a.x * S);
}
// multiply by a table twiddle (forward table value w); inverse uses conj(w)
template <bool INV> __device__ __forceinline__ cf twm(cf a, cf w) { return INV ? cmulc(a, w) : cmul(a, w); }

#define SP_C8 0.70710678118654752440f
#define SP_C16 0.92387953251128675613f
#define SP_S16 0.38268343236508977173f

// ---- in-register DFTs, natural-order output -------------------------------------------
template <bool INV> __device__ __forceinline__ void dft2(cf &a, cf &b) {
    cf t = a - b;
    a = a + b;
    b = t;
}

template <bool INV> __device__ __forceinline__ void dft4(cf &a, cf &b, cf &c, cf &d) {
    cf t0 = a + c, t1 = a - c, t2 = b + d, t3 = rot<INV>(b - d);
    a = t0 + t2;
    c = t0 - t2;
    b = t1 + t3;
    d = t1 - t3;
}

template <bool INV> __device__ __forceinline__ void dft8(cf (&x)[8]) {
    dft4<INV>(x[0], x[2], x[4], x[6]);   // E0..E3 in x0,x2,x4,x6
    dft4<INV>(x[1], x[3], x[5], x[7]);   // O0..O3 in x1,x3,x5,x7
    cf o1 = twc<INV>(x[3], SP_C8, SP_C8);
    cf o2 = rot<INV>(x[5]);
    cf o3 = twc<INV>(x[7], -SP_C8, SP_C8);
    cf e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6], o0 = x[1];
    x[0] = e0 + o0;
    x[4] = e0 - o0;
    x[1] = e1 + o1;
    x[5] = e1 - o1;
    x[2] = e2 + o2;
    x[6] = e2 - o2;
    x[3] = e3 + o3;
    x[7] = e3 - o3;
}

template <bool INV> __device__ __forceinline__ void dft16(cf (&x)[16]) {
    // n = 4a + b, k = c + 4d:  W16^{nk} = W4^{ac} W16^{bc} W4^{bd}
#pragma unroll
    for (int b = 0; b < 4; ++b) dft4<INV>(x[b], x[b + 4], x[b + 8], x[b + 12]);   // x[4c+b] = Y[b][c]
    x[5] = twc<INV>(x[5], SP_C16, SP_S16);      // b=1,c=1  W16^1
    x[9] = twc<INV>(x[9], SP_C8, SP_C8);        // b=1,c=2  W16^2
    x[13] = twc<INV>(x[13], SP_S16, SP_C16);    // b=1,c=3  W16^3
    x[6] = twc<INV>(x[6], SP_C8, SP_C8);        // b=2,c=1  W16^2
    x[10] = rot<INV>(x[10]);                    // b=2,c=2  W16^4
    x[14] = twc<INV>(x[14], -SP_C8, SP_C8);     // b=2,c=3  W16^6
    x[7] = twc<INV>(x[7], SP_S16, SP_C16);      // b=3,c=1  W16^3
    x[11] = twc<INV>(x[11], -SP_C8, SP_C8);     // b=3,c=2  W16^6
    x[15] = twc<INV>(x[15], -SP_C16, -SP_S16);  // b=3,c=3  W16^9
#pragma unroll
    for (int c = 0; c < 4; ++c) dft4<INV>(x[4 * c], x[4 * c + 1], x[4 * c + 2], x[4 * c + 3]);   // x[4c+d] = X[c+4d]
    // transpose the 4x4 index to natural order
    cf t;
#define SP_SWAP(i, j) t = x[i]; x[i] = x[j]; x[j] = t;
    SP_SWAP(1, 4) SP_SWAP(2, 8) SP_SWAP(3, 12) SP_SWAP(6, 9) SP_SWAP(7, 13) SP_SWAP(11, 14)
#undef SP_SWAP
}

// ---- radix-16 with FMA-fused ("scaled tangent") twiddles, forward only --------------------------------------------
// A twiddle w = exp(-i th) is applied as  w = g (1 - i tau),  g = cos th, tau = tan th:  the rotation  x (1 - i tau)
// costs 2 FMAs, and the real scale g rides along as a PENDING factor that the next radix-4 butterfly absorbs into its
// additions (a + g c is one FMA).  Normalising a butterfly by its first input's scale leaves three ratios per
// butterfly; the W16 constants between the two radix-4 stages are folded the same way.  Per twiddled radix-16 pass:
// 30 (rotate) + 64 + 16 + 64 = 174 VALU instead of 60 + 162; an untwiddled one 144 instead of 162.  cos th = 0 occurs
// only where the twiddle is exactly -i (or +i): g is clamped to 2^-40 in magnitude there (error 1e-12), tau ~ 1e12
// stays far from the float range for any data a float FFT could hold; near-zero g elsewhere is >= sin(2 pi/8192).
struct Tw16 {
    // f[0..14] tau = tan(th_s), s = 1..15
    // f[15..18] rb, f[19..22] rc, f[23..26] rd : stage 1, group b: g[b+4]/g[b], g[b+8]/g[b], g[b+12]/g[b+4]  (g[0] = 1)
    // f[27..30] s1, f[31..34] s2, f[35..38] s3 : stage 2, output c: g[1] K1c, g[2] K2c, g[3] K3c / (g[1] K1c)
    float f[40];
    static constexpr int TAU = 0, RB = 15, RC = 19, RD = 23, S1 = 27, S2 = 31, S3 = 35;
};

#define SP_T16 0.41421356237309504880f   // tan(pi/8)
#define SP_T316 2.41421356237309504880f  // tan(3 pi/8)

// forward radix-4 on (a, gb*b, gc*c, gd*d) with rb = gb, rc = gc, rd = gd/gb; 16 FMAs
__device__ __forceinline__ void dft4s(cf &a, cf &b, cf &c, cf &d, float rb, float rc, float rd) {
    const cf t0 = mk(fmaf(rc, c.x, a.x), fmaf(rc, c.y, a.y));
    const cf t1 = mk(fmaf(-rc, c.x, a.x), fmaf(-rc, c.y, a.y));
    const cf t2 = mk(fmaf(rd, d.x, b.x), fmaf(rd, d.y, b.y));
    const cf t3 = mk(fmaf(-rd, d.x, b.x), fmaf(-rd, d.y, b.y));
    a = mk(fmaf(rb, t2.x, t0.x), fmaf(rb, t2.y, t0.y));
    c = mk(fmaf(-rb, t2.x, t0.x), fmaf(-rb, t2.y, t0.y));
    b = mk(fmaf(rb, t3.y, t1.x), fmaf(-rb, t3.x, t1.y));     // t1 + rb (-i t3)
    d = mk(fmaf(-rb, t3.y, t1.x), fmaf(rb, t3.x, t1.y));
}
// x (1 - i t)
__device__ __forceinline__ cf rot_tan(cf a, float t) { return mk(fmaf(t, a.y, a.x), fmaf(-t, a.x, a.y)); }

// PREROT: the inputs arrive already rotated by (1 - i tau_s) -- the producing pass applied the rotation to its outputs before the
// exchange (wave-specialised pipeline: the rotations of pass p + 1 are work of role p); the pending scales g_s are absorbed here
// as always
template <bool TWD, bool PREROT = false> __device__ __forceinline__ void dft16s(cf (&x)[16], const Tw16 &w) {
    if constexpr (TWD) {
        if constexpr (!PREROT) {
#pragma unroll
            for (int s = 1; s < 16; ++s) x[s] = rot_tan(x[s], w.f[Tw16::TAU + s - 1]);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) dft4s(x[b], x[b + 4], x[b + 8], x[b + 12], w.f[Tw16::RB + b], w.f[Tw16::RC + b], w.f[Tw16::RD + b]);
    } else {
#pragma unroll
        for (int b = 0; b < 4; ++b) dft4<false>(x[b], x[b + 4], x[b + 8], x[b + 12]);
    }
    // x[4c+b] = Y[b][c] (pending scale g[b]); rotate by W16^{bc} / K_bc
    x[5] = rot_tan(x[5], SP_T16);                          // W16^1 = C16 (1 - i tan(pi/8))
    x[9] = mk(x[9].x + x[9].y, x[9].y - x[9].x);           // W16^2 = C8 (1 - i)
    x[13] = rot_tan(x[13], SP_T316);                       // W16^3 = S16 (1 - i tan(3pi/8))
    x[6] = mk(x[6].x + x[6].y, x[6].y - x[6].x);           // W16^2
    x[10] = mk(x[10].y, -x[10].x);                         // W16^4 = -i
    x[14] = mk(x[14].x - x[14].y, x[14].y + x[14].x);      // W16^6 = -C8 (1 + i)
    x[7] = rot_tan(x[7], SP_T316);                         // W16^3
    x[11] = mk(x[11].x - x[11].y, x[11].y + x[11].x);      // W16^6
    x[15] = rot_tan(x[15], SP_T16);                        // W16^9 = -C16 (1 - i tan(pi/8))
    if constexpr (TWD) {
#pragma unroll
        for (int c = 0; c < 4; ++c) dft4s(x[4 * c], x[4 * c + 1], x[4 * c + 2], x[4 * c + 3], w.f[Tw16::S1 + c], w.f[Tw16::S2 + c], w.f[Tw16::S3 + c]);
    } else {
        dft4<false>(x[0], x[1], x[2], x[3]);
        dft4s(x[4], x[5], x[6], x[7], SP_C16, SP_C8, SP_S16 / SP_C16);
        dft4s(x[8], x[9], x[10], x[11], SP_C8, 1.f, -1.f);              // K3c/K1c = -C8/C8
        dft4s(x[12], x[13], x[14], x[15], SP_S16, -SP_C8, -SP_C16 / SP_S16);
    }
    cf t;
#define SP_SWAP(i, j) t = x[i]; x[i] = x[j]; x[j] = t;
    SP_SWAP(1, 4) SP_SWAP(2, 8) SP_SWAP(3, 12) SP_SWAP(6, 9) SP_SWAP(7, 13) SP_SWAP(11, 14)
#undef SP_SWAP
}

// The same radix-16, but every finished output is handed to `store(k, value)` (k = natural-order output index) right
// after the radix-4 butterfly that produced it, and the instruction scheduler may not move anything across the end of a
// butterfly: the 16 LDS stores of a Stockham scatter are then spread over the last 64 VALU instructions of the pass
// instead of forming one burst behind it (SP_EARLY_SCATTER; the burst is where the waves queue on the LDS pipe:
// SQ_WAIT_INST_LDS was 16 % of the wave cycles of the metric kernel).
template <bool TWD, class Store, bool PREROT = false> __device__ __forceinline__ void dft16s_es(cf (&x)[16], const Tw16 &w, Store store) {
    if constexpr (TWD) {
        if constexpr (!PREROT) {
#pragma unroll
            for (int s = 1; s < 16; ++s) x[s] = rot_tan(x[s], w.f[Tw16::TAU + s - 1]);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) dft4s(x[b], x[b + 4], x[b + 8], x[b + 12], w.f[Tw16::RB + b], w.f[Tw16::RC + b], w.f[Tw16::RD + b]);
    } else {
#pragma unroll
        for (int b = 0; b < 4; ++b) dft4<false>(x[b], x[b + 4], x[b + 8], x[b + 12]);
    }
    x[5] = rot_tan(x[5], SP_T16);
    x[9] = mk(x[9].x + x[9].y, x[9].y - x[9].x);
    x[13] = rot_tan(x[13], SP_T316);
    x[6] = mk(x[6].x + x[6].y, x[6].y - x[6].x);
    x[10] = mk(x[10].y, -x[10].x);
    x[14] = mk(x[14].x - x[14].y, x[14].y + x[14].x);
    x[7] = rot_tan(x[7], SP_T316);
    x[11] = mk(x[11].x - x[11].y, x[11].y + x[11].x);
    x[15] = rot_tan(x[15], SP_T16);
    __builtin_amdgcn_sched_barrier(0);
#define SP_ES_OUT(c)                                                                                                 \
    store(c, x[4 * c]);                                                                                              \
    store(c + 4, x[4 * c + 1]);                                                                                      \
    store(c + 8, x[4 * c + 2]);                                                                                      \
    store(c + 12, x[4 * c + 3]);                                                                                     \
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TWD) {
        dft4s(x[0], x[1], x[2], x[3], w.f[Tw16::S1 + 0], w.f[Tw16::S2 + 0], w.f[Tw16::S3 + 0]);
        SP_ES_OUT(0)
        dft4s(x[4], x[5], x[6], x[7], w.f[Tw16::S1 + 1], w.f[Tw16::S2 + 1], w.f[Tw16::S3 + 1]);
        SP_ES_OUT(1)
        dft4s(x[8], x[9], x[10], x[11], w.f[Tw16::S1 + 2], w.f[Tw16::S2 + 2], w.f[Tw16::S3 + 2]);
        SP_ES_OUT(2)
        dft4s(x[12], x[13], x[14], x[15], w.f[Tw16::S1 + 3], w.f[Tw16::S2 + 3], w.f[Tw16::S3 + 3]);
        SP_ES_OUT(3)
    } else {
        dft4<false>(x[0], x[1], x[2], x[3]);
        SP_ES_OUT(0)
        dft4s(x[4], x[5], x[6], x[7], SP_C16, SP_C8, SP_S16 / SP_C16);
        SP_ES_OUT(1)
        dft4s(x[8], x[9], x[10], x[11], SP_C8, 1.f, -1.f);
        SP_ES_OUT(2)
        dft4s(x[12], x[13], x[14], x[15], SP_S16, -SP_C8, -SP_C16 / SP_S16);
        SP_ES_OUT(3)
    }
#undef SP_ES_OUT
}

// The untwiddled first pass with the WINDOW folded into its first radix-4 stage: the inputs are the raw samples r[t] and
// the window values w[t]; instead of 32 multiplications v = w r followed by 16 additions per butterfly, the pair sums are
// formed as a' = w_a r_a, t0 = fma(w_c, r_c, a'), t1 = fma(-w_c, r_c, a') (and b', t2, t3 alike): 20 instructions per
// butterfly instead of 8 + 16 -- 16 VALU instructions per frame and thread less (585 instead of 601 at 4096 points).
__device__ __forceinline__ void dft4w(cf &a, cf &b, cf &c, cf &d, float wa, float wb, float wc, float wd) {
    const cf pa = mk(wa * a.x, wa * a.y), pb = mk(wb * b.x, wb * b.y);
    const cf t0 = mk(fmaf(wc, c.x, pa.x), fmaf(wc, c.y, pa.y));
    const cf t1 = mk(fmaf(-wc, c.x, pa.x), fmaf(-wc, c.y, pa.y));
    const cf t2 = mk(fmaf(wd, d.x, pb.x), fmaf(wd, d.y, pb.y));
    const cf t3 = mk(fmaf(-wd, d.x, pb.x), fmaf(-wd, d.y, pb.y));
    a = t0 + t2;
    c = t0 - t2;
    b = mk(t1.x + t3.y, t1.y - t3.x);           // t1 - i t3
    d = mk(t1.x - t3.y, t1.y + t3.x);
}
struct NoMid {
    __device__ __forceinline__ void operator()(int) const {}
};
// mid(b), b = 0..2: called behind the b-th first-stage butterfly, pinned there (the pipeline's front role issues its spread sample
// loads at these points -- a quarter of a pass apart -- instead of behind the last stage's stores)
template <class Store, class Mid = NoMid> __device__ __forceinline__ void dft16s_es_win(cf (&x)[16], const float (&w)[16], Store store, Mid mid = Mid()) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        dft4w(x[b], x[b + 4], x[b + 8], x[b + 12], w[b], w[b + 4], w[b + 8], w[b + 12]);
        if constexpr (!std::is_same<Mid, NoMid>::value) {
            if (b < 3) {
                __builtin_amdgcn_sched_barrier(0);
                mid(b);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    x[5] = rot_tan(x[5], SP_T16);
    x[9] = mk(x[9].x + x[9].y, x[9].y - x[9].x);
    x[13] = rot_tan(x[13], SP_T316);
    x[6] = mk(x[6].x + x[6].y, x[6].y - x[6].x);
    x[10] = mk(x[10].y, -x[10].x);
    x[14] = mk(x[14].x - x[14].y, x[14].y + x[14].x);
    x[7] = rot_tan(x[7], SP_T316);
    x[11] = mk(x[11].x - x[11].y, x[11].y + x[11].x);
    x[15] = rot_tan(x[15], SP_T16);
    __builtin_amdgcn_sched_barrier(0);
#define SP_ES_OUT(c)                                                                                                 \
    store(c, x[4 * c]);                                                                                              \
    store(c + 4, x[4 * c + 1]);                                                                                      \
    store(c + 8, x[4 * c + 2]);                                                                                      \
    store(c + 12, x[4 * c + 3]);                                                                                     \
    __builtin_amdgcn_sched_barrier(0);
    dft4<false>(x[0], x[1], x[2], x[3]);
    SP_ES_OUT(0)
    dft4s(x[4], x[5], x[6], x[7], SP_C16, SP_C8, SP_S16 / SP_C16);
    SP_ES_OUT(1)
    dft4s(x[8], x[9], x[10], x[11], SP_C8, 1.f, -1.f);
    SP_ES_OUT(2)
    dft4s(x[12], x[13], x[14], x[15], SP_S16, -SP_C8, -SP_C16 / SP_S16);
    SP_ES_OUT(3)
#undef SP_ES_OUT
}

// ---- the same butterflies in packed fp32 (v_pk_fma_f32 / v_pk_add_f32), SP_PACKED=1 ---------------------------------
// A complex value is one 64-bit register pair and every butterfly line is ONE packed instruction; the swap of re/im,
// the broadcast of a real scale out of a register pair and the signs ride in the op_sel / neg modifiers (inline asm:
// hipcc folds whole-vector negations and swizzles but materialises mixed-sign operands).  87 (twiddled) / 72 (first
// pass) instructions per radix-16 instead of 174 / 144; the SIMD's packed rate is the same flops per cycle as the
// scalar one, but one wave alone can reach it (a wave issues at most one VALU instruction per ~5.5 cycles whatever the
// instruction does, tools/ubench/valu_rate.hip).  Bit-identical results, all tests pass -- and MEASURED NO FASTER:
// Welch 0.665 vs 0.640 ms, FFT / STFT / FIR equal within 1-7 % (profiles/r01_ubench.txt notes).  Kept as an option;
// the scalar form is the default.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f to_v2f(cf a) { return (v2f){a.x, a.y}; }
__device__ __forceinline__ cf to_cf(v2f a) { return mk(a.x, a.y); }
// a + (NEG ? -1 : 1) * s[H] * c          (s[H]: half H of the pair s, broadcast)
template <int H, int NEG> __device__ __forceinline__ v2f pk_sfma(v2f s, v2f c, v2f a) {
    v2f d;
    if constexpr (H == 0 && NEG == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(d) : "v"(s), "v"(c), "v"(a));
    if constexpr (H == 1 && NEG == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(s), "v"(c), "v"(a));
    if constexpr (H == 0 && NEG == 1) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(s), "v"(c), "v"(a));
    if constexpr (H == 1 && NEG == 1) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(s), "v"(c), "v"(a));
    return d;
}
// a + (NEG ? -1 : 1) * s[H] * (-i c)  =  (a.x +- s c.y,  a.y -+ s c.x)
template <int H, int NEG> __device__ __forceinline__ v2f pk_srot(v2f s, v2f c, v2f a) {
    v2f d;
    if constexpr (H == 0 && NEG == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(d) : "v"(s), "v"(c), "v"(a));
    if constexpr (H == 1 && NEG == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "=v"(d) : "v"(s), "v"(c), "v"(a));
    if constexpr (H == 0 && NEG == 1) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(s), "v"(c), "v"(a));
    if constexpr (H == 1 && NEG == 1) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(s), "v"(c), "v"(a));
    return d;
}
// a + (NEG ? -1 : 1) * (-i c)
template <int NEG> __device__ __forceinline__ v2f pk_rot(v2f c, v2f a) {
    v2f d;
    if constexpr (NEG == 0) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(c));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(c));
    return d;
}
// forward radix-4 on (a, gb b, gc c, gd d); the scales are halves HB/HC/HD of the pairs pb/pc/pd; rd = gd/gb.  8 instr.
template <int HB, int HC, int HD> __device__ __forceinline__ void dft4p(v2f &a, v2f &b, v2f &c, v2f &d, v2f pb, v2f pc, v2f pd) {
    const v2f t0 = pk_sfma<HC, 0>(pc, c, a), t1 = pk_sfma<HC, 1>(pc, c, a);
    const v2f t2 = pk_sfma<HD, 0>(pd, d, b), t3 = pk_sfma<HD, 1>(pd, d, b);
    a = pk_sfma<HB, 0>(pb, t2, t0);
    c = pk_sfma<HB, 1>(pb, t2, t0);
    b = pk_srot<HB, 0>(pb, t3, t1);
    d = pk_srot<HB, 1>(pb, t3, t1);
}
__device__ __forceinline__ void dft4p_plain(v2f &a, v2f &b, v2f &c, v2f &d) {
    const v2f t0 = a + c, t1 = a - c, t2 = b + d, t3 = b - d;
    a = t0 + t2;
    c = t0 - t2;
    b = pk_rot<0>(t3, t1);
    d = pk_rot<1>(t3, t1);
}
// packed constants of a radix-16 pass: p[j >> 1] half j & 1 holds Tw16::f[j]
struct Tw16p {
    v2f p[20];
};
__device__ __forceinline__ void pack_tw16(Tw16p &o, const Tw16 &w) {
#pragma unroll
    for (int i = 0; i < 20; ++i) o.p[i] = (v2f){w.f[2 * i], w.f[2 * i + 1]};
}
// K0 = (C16, C8)  K1 = (S16/C16, S16)  K2 = (-C8, -C16/S16)  K3 = (T16, T316): the W16 constants, kept in VGPR pairs
// (a VALU source that is an SGPR issues at half rate)
struct K16p {
    v2f k0, k1, k2, k3;
};
__device__ __forceinline__ K16p make_k16p() {
    K16p k;
    k.k0 = (v2f){SP_C16, SP_C8};
    k.k1 = (v2f){SP_S16 / SP_C16, SP_S16};
    k.k2 = (v2f){-SP_C8, -SP_C16 / SP_S16};
    k.k3 = (v2f){SP_T16, SP_T316};
    asm volatile("" : "+v"(k.k0), "+v"(k.k1), "+v"(k.k2), "+v"(k.k3));
    return k;
}
#define SP_TWP(w, j) (w).p[(j) >> 1]
#define SP_TWH(j) ((j) & 1)
template <bool TWD> __device__ __forceinline__ void dft16p(cf (&xc)[16], const Tw16p &w, const K16p &k) {
    v2f x[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) x[s] = to_v2f(xc[s]);
    if constexpr (TWD) {
        // x (1 - i tau) = x + tau (-i x)
#define SP_RT(s) x[s] = pk_srot<SP_TWH(Tw16::TAU + s - 1), 0>(SP_TWP(w, Tw16::TAU + s - 1), x[s], x[s]);
        SP_RT(1) SP_RT(2) SP_RT(3) SP_RT(4) SP_RT(5) SP_RT(6) SP_RT(7) SP_RT(8) SP_RT(9) SP_RT(10) SP_RT(11) SP_RT(12)
        SP_RT(13) SP_RT(14) SP_RT(15)
#undef SP_RT
#define SP_G1(b)                                                                                                     \
    dft4p<SP_TWH(Tw16::RB + b), SP_TWH(Tw16::RC + b), SP_TWH(Tw16::RD + b)>(x[b], x[b + 4], x[b + 8], x[b + 12],     \
                                                                           SP_TWP(w, Tw16::RB + b), SP_TWP(w, Tw16::RC + b), \
                                                                           SP_TWP(w, Tw16::RD + b));
        SP_G1(0) SP_G1(1) SP_G1(2) SP_G1(3)
#undef SP_G1
    } else {
#pragma unroll
        for (int b = 0; b < 4; ++b) dft4p_plain(x[b], x[b + 4], x[b + 8], x[b + 12]);
    }
    // x[4c+b] = Y[b][c] (pending scale g[b]); rotate by W16^{bc} / K_bc
    x[5] = pk_srot<0, 0>(k.k3, x[5], x[5]);        // W16^1 = C16 (1 - i tan(pi/8))
    x[9] = pk_rot<0>(x[9], x[9]);                  // W16^2 = C8 (1 - i)
    x[13] = pk_srot<1, 0>(k.k3, x[13], x[13]);     // W16^3 = S16 (1 - i tan(3pi/8))
    x[6] = pk_rot<0>(x[6], x[6]);                  // W16^2
    x[10] = pk_rot<0>(x[10], (v2f){0.f, 0.f});     // W16^4 = -i
    x[14] = pk_rot<1>(x[14], x[14]);               // W16^6 = -C8 (1 + i)
    x[7] = pk_srot<1, 0>(k.k3, x[7], x[7]);        // W16^3
    x[11] = pk_rot<1>(x[11], x[11]);               // W16^6
    x[15] = pk_srot<0, 0>(k.k3, x[15], x[15]);     // W16^9 = -C16 (1 - i tan(pi/8))
    if constexpr (TWD) {
#define SP_G2(c)                                                                                                     \
    dft4p<SP_TWH(Tw16::S1 + c), SP_TWH(Tw16::S2 + c), SP_TWH(Tw16::S3 + c)>(x[4 * c], x[4 * c + 1], x[4 * c + 2], x[4 * c + 3], \
                                                                           SP_TWP(w, Tw16::S1 + c), SP_TWP(w, Tw16::S2 + c), \
                                                                           SP_TWP(w, Tw16::S3 + c));
        SP_G2(0) SP_G2(1) SP_G2(2) SP_G2(3)
#undef SP_G2
    } else {
        dft4p_plain(x[0], x[1], x[2], x[3]);
        dft4p<0, 1, 0>(x[4], x[5], x[6], x[7], k.k0, k.k0, k.k1);             // rb = C16, rc = C8, rd = S16/C16
        {
            // rb = C8, rc = 1, rd = -1
            const v2f t0 = x[8] + x[10], t1 = x[8] - x[10], t2 = x[9] - x[11], t3 = x[9] + x[11];
            x[8] = pk_sfma<1, 0>(k.k0, t2, t0);
            x[10] = pk_sfma<1, 1>(k.k0, t2, t0);
            x[9] = pk_srot<1, 0>(k.k0, t3, t1);
            x[11] = pk_srot<1, 1>(k.k0, t3, t1);
        }
        dft4p<1, 0, 1>(x[12], x[13], x[14], x[15], k.k1, k.k2, k.k2);         // rb = S16, rc = -C8, rd = -C16/S16
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) xc[s] = to_cf(x[s]);
    cf t;
#define SP_SWAP(i, j) t = xc[i]; xc[i] = xc[j]; xc[j] = t;
    SP_SWAP(1, 4) SP_SWAP(2, 8) SP_SWAP(3, 12) SP_SWAP(6, 9) SP_SWAP(7, 13) SP_SWAP(11, 14)
#undef SP_SWAP
}

// per-thread constants of a twiddled radix-16 pass from the forward twiddles wv[s-1] = exp(-i s phi), s = 1..15
__device__ __forceinline__ void make_tw16(Tw16 &w, const cf (&wv)[15]) {
    float g[16];
    g[0] = 1.f;
    w.f[39] = 0.f;
#pragma unroll
    for (int s = 1; s < 16; ++s) {
        float c = wv[s - 1].x;
        if (fabsf(c) < 9.094947e-13f) c = c < 0.f ? -9.094947e-13f : 9.094947e-13f;   // 2^-40
        g[s] = c;
        w.f[Tw16::TAU + s - 1] = -wv[s - 1].y / c;
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        w.f[Tw16::RB + b] = g[b + 4] / g[b];
        w.f[Tw16::RC + b] = g[b + 8] / g[b];
        w.f[Tw16::RD + b] = g[b + 12] / g[b + 4];
    }
    const float K1[4] = {1.f, SP_C16, SP_C8, SP_S16}, K2[4] = {1.f, SP_C8, 1.f, -SP_C8}, K3[4] = {1.f, SP_S16, -SP_C8, -SP_C16};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        w.f[Tw16::S1 + c] = g[1] * K1[c];
        w.f[Tw16::S2 + c] = g[2] * K2[c];
        w.f[Tw16::S3 + c] = (g[3] * K3[c]) / (g[1] * K1[c]);
    }
}

// SP_ABLATE (diagnostic builds only, results wrong): bit 0 = skip the radix-16 butterflies, bit 1 = skip the LDS
// exchanges (and their barriers), bit 2 = skip the inter-pass twiddles, bit 3 = no global loads in the carry loop, bit 4 = the loop's loads all hit L2
// SP_PACKED=1: radix-16 butterflies in packed fp32 (v_pk_*_f32); 0: the scalar FMA form
#ifndef SP_PACKED
#define SP_PACKED 0
#endif
// SP_DIAG_SHARETW (diagnostic, results wrong): every twiddled radix-16 pass uses the first pass's constants
#ifndef SP_DIAG_SHARETW
#define SP_DIAG_SHARETW 0
#endif
#ifndef SP_ABLATE
#define SP_ABLATE 0
#endif
// SP_READ_B64=1: the unit-stride gathers are issued as single ds_read_b64 (volatile LDS loads, which hipcc does not merge
// into ds_read2_b64): 2 LDS cycles per 8 bytes per wave instead of 8 per 16 (MI355X_MICROARCH.md, LDS table); the first
// exchange image then takes row pitch T+2 (32-lane groups over 64 banks) instead of T+1
#ifndef SP_READ_B64
#define SP_READ_B64 0
#endif
// SP_EARLY_SCATTER=1 (callers with two exchange images only): see dft16s_es
#ifndef SP_EARLY_SCATTER
#define SP_EARLY_SCATTER 0
#endif
typedef float sp_f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cf lds_load_single(const cf *p) {
#if SP_READ_B64
    // explicit LDS address space: a volatile access through a generic pointer is not narrowed to ds_read by hipcc
    typedef const volatile __attribute__((address_space(3))) sp_f2v *lds_cvp;
    const sp_f2v r = *(lds_cvp)(p);
    return mk(r.x, r.y);
#else
    return *p;
#endif
}
template <int RDX, bool INV> __device__ __forceinline__ void dftR(cf (&x)[RDX]) {
    if constexpr (RDX == 2) dft2<INV>(x[0], x[1]);
    else if constexpr (RDX == 4) dft4<INV>(x[0], x[1], x[2], x[3]);
    else if constexpr (RDX == 8) dft8<INV>(x);
    else if constexpr (RDX == 16) dft16<INV>(x);
}

// ---- plan ------------------------------------------------------------------------------
constexpr int ilog2c(int n) { return n <= 1 ? 0 : 1 + ilog2c(n >> 1); }

template <int N> struct FftPlan {
    static_assert(N >= 2 && (N & (N - 1)) == 0, "power of two");
    static constexpr int LOG2N = ilog2c(N);
    static constexpr int R = N < 16 ? N : 16;                  // points per thread
    static constexpr int T = N / R;                            // threads per transform
    static constexpr int NP16 = N < 16 ? 0 : LOG2N / 4;        // radix-16 passes
    static constexpr int REM = N < 16 ? N : (1 << (LOG2N % 4));   // last-pass radix (1 = none)
    static constexpr int NP = N < 16 ? 1 : NP16 + (REM > 1 ? 1 : 0);
    static constexpr int radix(int p) { return N < 16 ? N : (p < NP16 ? 16 : REM); }
    static constexpr int ns(int p) { int s = 1; for (int i = 0; i < p; ++i) s *= radix(i); return s; }
    // per-thread twiddle count: passes p>=1, (R/r) butterflies x (r-1) factors
    static constexpr int ntw_before(int p) { int c = 0; for (int i = 1; i < p; ++i) c += (R / radix(i)) * (radix(i) - 1); return c; }
    static constexpr int NTW = ntw_before(NP);
    // first-exchange row pitch (complex).  hipcc merges the unit-stride reads into ds_read2_b64 (16-lane groups,
    // bank = dword mod 32): lane stride must be == 2 dwords mod 32 -> pitch == 1 mod 16.  (T+2 would suit plain
    // ds_read_b64 -- 32-lane groups, 64 banks -- and costs a 2-way conflict on every read2: measured 256 LDS
    // conflict cycles per 4096-point frame.)
    static constexpr int PITCH1 = SP_READ_B64 ? T + 2 : T + 1;
    static constexpr int LDS_ELEMS = NP > 1 ? (R * PITCH1 > N ? R * PITCH1 : N) : 0;
};

// ---- the workgroup FFT ------------------------------------------------------------------
// Forward transforms only (inverses are conj(fft(conj(.))) at the call sites).
// TW1LDS: the constants of the FIRST twiddled radix-16 pass are not kept in registers but re-read every transform from a
// 16-row table in LDS (`tw1`: row = tid % 16, SP_TW1_PITCH floats apart) -- only 16 distinct sets exist for that pass; frees
// 40 VGPRs per thread (the occupancy experiment of the carry kernel, SP_CARRY_W3)
#define SP_TW1_PITCH 44      /* 40 floats + 4: rows start 44 dwords apart -> the 16 rows of a b128 read hit 16 distinct bank quads */
// RM: the first exchange image holds every thread's 16 pass-0 outputs side by side, [thread][16] at a pitch of RM_PITCH, the
// four outputs of a final radix-4 butterfly (k = c, c+4, c+8, c+12) adjacent: two 16-byte writes instead of four 8-byte ones
// (the [k][thread] image at pitch T + 1 takes 16 single ds_write_b64: its stride fits neither write2 form), and the unit-
// stride gather of the next pass strides 16 RM_PITCH = 5 x 64 elements, a ds_read2st64_b64 per two elements as before.
template <int N, bool TW1LDS = false, bool RM = false> struct WgFft {
    using PL = FftPlan<N>;
    static constexpr int R = PL::R, T = PL::T, NP = PL::NP;
    static constexpr int RM_PITCH = 20;
    static constexpr int IMG0 = RM ? T * RM_PITCH : PL::LDS_ELEMS;      // elements of a first-exchange image
    static_assert(!RM || (PL::radix(0) == 16 && R == 16 && (T % 16) == 0), "RM: radix-16 first pass");
    static constexpr int N16 = PL::NP16 > 1 ? PL::NP16 - 1 : 0;          // twiddled radix-16 passes
    static constexpr int NTR = (N >= 16 && PL::REM > 1) ? (R / PL::REM) * (PL::REM - 1) : 0;   // remainder-pass twiddles
#if SP_PACKED
    Tw16p t16[N16 > 0 ? N16 : 1];
    K16p k16;
#else
    Tw16 t16[N16 > 0 ? N16 : 1];
#endif
    cf twr[NTR > 0 ? NTR : 1];
    const float *tw1 = nullptr;

    // TW1LDS: write this thread's pass-1 constants into row tid % 16 of the LDS table (threads 0..15 cover all rows); call
    // after load_twiddles, followed by a barrier
    __device__ __forceinline__ void publish_tw1(float *table, int tid) {
#if !SP_PACKED
        if constexpr (TW1LDS && N16 > 0) {
            if (tid < 16) {
#pragma unroll
                for (int j = 0; j < 40; ++j) table[tid * SP_TW1_PITCH + j] = t16[0].f[j];
            }
            tw1 = table;
        }
#endif
    }

    // table[m] = exp(-2 pi i m / N), m = 0..N-1
    __device__ __forceinline__ void load_twiddles(const cf *__restrict__ table, int tid) {
#if SP_PACKED
        k16 = make_k16p();
#endif
        load_tw<1>(table, tid);
    }

    template <int P> __device__ __forceinline__ void load_tw(const cf *__restrict__ table, int tid) {
        if constexpr (P < NP) {
            constexpr int r = PL::radix(P), NS = PL::ns(P), NB = R / r;
            if constexpr (r == 16) {
                const int e = (tid % NS) * (N / (NS * 16));
                cf wv[15];
#pragma unroll
                for (int s = 1; s < 16; ++s) wv[s - 1] = table[e * s];
#if SP_PACKED
                Tw16 w1;
                make_tw16(w1, wv);
                pack_tw16(t16[P - 1], w1);
#else
                make_tw16(t16[P - 1], wv);
#endif
            } else {
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int q = tid + T * u;
                    const int e = (q % NS) * (N / (NS * r));
#pragma unroll
                    for (int s = 1; s < r; ++s) twr[u * (r - 1) + (s - 1)] = table[e * s];
                }
            }
            load_tw<P + 1>(table, tid);
        }
    }

    // constants of ONE twiddled radix-16 pass P >= 1 (wave-specialised callers keep only their own pass's set)
    template <int P> __device__ __forceinline__ void load_tw_one(const cf *__restrict__ table, int tid) {
        static_assert(P >= 1 && P < NP && PL::radix(P) == 16, "a twiddled radix-16 pass");
        constexpr int NS = PL::ns(P);
        const int e = (tid % NS) * (N / (NS * 16));
        cf wv[15];
#pragma unroll
        for (int s = 1; s < 16; ++s) wv[s - 1] = table[e * s];
#if SP_PACKED
        Tw16 w1;
        make_tw16(w1, wv);
        pack_tw16(t16[P - 1], w1);
        k16 = make_k16p();
#else
        make_tw16(t16[P - 1], wv);
#endif
    }

    // physical LDS index of logical element i for exchange number E (0 = first)
    template <int E> static __device__ __forceinline__ int phys(int i) {
        if constexpr (E == 0 && RM) return (i / 16) * RM_PITCH + (i % 4) * 4 + (i % 16) / 4;
        else if constexpr (E == 0) return (i % PL::radix(0)) * PL::PITCH1 + i / PL::radix(0);
        else return i;
    }

    // ---- the three pieces of a pass, usable on their own (software-pipelined callers) -------------------------
    // butterflies of pass P, in place: v[u + s*NB] is input s and then output s of butterfly u
    template <int P> __device__ __forceinline__ void bfly(cf (&v)[R], int tid) const {
        constexpr int r = PL::radix(P), NB = R / r;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            cf x[r];
#pragma unroll
            for (int s = 0; s < r; ++s) x[s] = v[u + s * NB];
            if constexpr (r == 16) {
                if constexpr (!(SP_ABLATE & 1)) {
#if SP_PACKED
                    if constexpr (P > 0 && !(SP_ABLATE & 4)) dft16p<true>(x, t16[SP_DIAG_SHARETW ? 0 : P - 1], k16);
                    else dft16p<false>(x, t16[0], k16);
#else
                    if constexpr (TW1LDS && P == 1 && !(SP_ABLATE & 4)) {
                        Tw16 wl;
                        const float4 *row = reinterpret_cast<const float4 *>(tw1 + (tid % 16) * SP_TW1_PITCH);
#pragma unroll
                        for (int j = 0; j < 10; ++j) {
                            const float4 q = row[j];
                            wl.f[4 * j] = q.x;
                            wl.f[4 * j + 1] = q.y;
                            wl.f[4 * j + 2] = q.z;
                            wl.f[4 * j + 3] = q.w;
                        }
                        dft16s<true>(x, wl);
                    } else if constexpr (P > 0 && !(SP_ABLATE & 4)) {
                        dft16s<true>(x, t16[SP_DIAG_SHARETW ? 0 : P - 1]);
                    } else {
                        dft16s<false>(x, t16[0]);
                    }
#endif
                }
            } else {
                if constexpr (P > 0 && !(SP_ABLATE & 4)) {
#pragma unroll
                    for (int s = 1; s < r; ++s) x[s] = twm<false>(x[s], twr[u * (r - 1) + (s - 1)]);
                }
                dftR<r, false>(x);
            }
#pragma unroll
            for (int s = 0; s < r; ++s) v[u + s * NB] = x[s];
        }
    }
    // Stockham scatter of pass P's outputs into exchange image P
    template <int P> __device__ __forceinline__ void scatter(const cf (&v)[R], cf *lds, int tid) const {
        constexpr int r = PL::radix(P), NS = PL::ns(P), NB = R / r;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int q = tid + T * u;
            const int base = (q / NS) * (NS * r) + (q % NS);
#pragma unroll
            for (int s = 0; s < r; ++s) lds[phys<P>(base + s * NS)] = v[u + s * NB];
        }
    }
    // unit-stride gather from exchange image P: v[t] = element tid + T*t
    template <int P> __device__ __forceinline__ void gather(cf (&v)[R], const cf *lds, int tid) const {
        if constexpr (P == 0 && RM) {
            // i = tid + T*t  ->  (tid/16 + (T/16) t) * RM_PITCH + pos(tid % 16)
            const int b = (tid / 16) * RM_PITCH + (tid % 4) * 4 + (tid % 16) / 4;
#pragma unroll
            for (int t = 0; t < R; ++t) v[t] = lds_load_single(&lds[b + (T / 16) * RM_PITCH * t]);
        } else if constexpr (P == 0 && (T % 16) == 0 && PL::radix(0) == 16) {
            // i = tid + T*t  ->  (i%16)*PITCH1 + i/16 = (tid%16)*PITCH1 + tid/16 + (T/16)*t
            const int b = (tid % 16) * PL::PITCH1 + tid / 16;
#pragma unroll
            for (int t = 0; t < R; ++t) v[t] = lds_load_single(&lds[b + (T / 16) * t]);
        } else {
#pragma unroll
            for (int t = 0; t < R; ++t) v[t] = lds_load_single(&lds[phys<P>(tid + T * t)]);
        }
    }

    // v[t] <-> element tid + T*t on entry and exit.  lds0/lds1: two exchange images (may be equal,
    // SINGLE=true, then a barrier also precedes every write).
    // WL (wave-local): the T threads of a transform are lanes of ONE wave (T <= 64, row-mapped kernels: thread % T): the
    // exchange needs no workgroup barrier -- LDS operations of a wave execute in order, so its gather sees its scatter -- and
    // the workgroup's other transforms (other waves) are not held up at an s_barrier for it
    template <bool SINGLE, bool WL = false> __device__ __forceinline__ void run(cf (&v)[R], cf *lds0, cf *lds1, int tid) const {
        static_assert(!WL || T <= 64, "wave-local exchange: one wave holds the whole transform");
        pass<0, SINGLE, WL>(v, lds0, lds1, tid);
    }
    template <bool WL> static __device__ __forceinline__ void xsync() {
        if constexpr (WL) __builtin_amdgcn_wave_barrier();
        else __syncthreads();
    }

    // butterflies of pass P with the Stockham scatter folded in (radix 16, one butterfly per thread, scalar form)
    template <int P> __device__ __forceinline__ void bfly_scatter(cf (&v)[R], cf *lds, int tid) const {
        constexpr int NS = PL::ns(P);
        const int base = (tid / NS) * (NS * 16) + (tid % NS);
        auto store = [&](int k, cf val) __attribute__((always_inline)) { lds[phys<P>(base + k * NS)] = val; };
        if constexpr (P > 0) dft16s_es<true>(v, t16[P - 1], store);
        else dft16s_es<false>(v, t16[0], store);
    }

    // tau_out[k], k = 0..15: the rotation (1 - i tau) that pass P + 1 applies to the element this thread's pass-P output k becomes
    // (Stockham position p = (tid / NS) 16 NS + tid % NS + k NS -> consumer thread p % T, slot p / T; its twiddle is table[e s]
    // with e = (thread % NS') (N / (16 NS')), NS' = 16 NS).  tau as make_tw16 forms it.
    template <int P> __device__ __forceinline__ void load_tau_out(const cf *__restrict__ table, int tid, float (&tau)[16]) const {
        static_assert(P + 1 < NP && PL::radix(P) == 16 && PL::radix(P + 1) == 16 && R == 16, "between two radix-16 passes");
        constexpr int NS = PL::ns(P), NS2 = PL::ns(P + 1);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int p = (tid / NS) * (NS * 16) + (tid % NS) + k * NS;
            const int tc = p % T, sc = p / T;
            const int e = (tc % NS2) * (N / (NS2 * 16));
            const cf w = table[(e * sc) & (N - 1)];
            float c = w.x;
            if (fabsf(c) < 9.094947e-13f) c = c < 0.f ? -9.094947e-13f : 9.094947e-13f;   // 2^-40, as make_tw16
            tau[k] = sc == 0 ? 0.f : -w.y / c;
        }
    }
    // butterflies of pass P on inputs that arrive pre-rotated (PREROT), outputs rotated for pass P + 1 and scattered
    template <int P, bool PREROT> __device__ __forceinline__ void bfly_scatter_rot(cf (&v)[R], const float (&tau)[16], cf *lds, int tid) const {
        constexpr int NS = PL::ns(P);
        const int base = (tid / NS) * (NS * 16) + (tid % NS);
        auto store = [&](int k, cf val) __attribute__((always_inline)) { lds[phys<P>(base + k * NS)] = rot_tan(val, tau[k]); };
        if constexpr (P > 0) dft16s_es<true, decltype(store), PREROT>(v, t16[P - 1], store);
        else dft16s_es<false>(v, t16[0], store);
    }
    __device__ __forceinline__ void bfly_scatter_win_rot(cf (&v)[R], const float (&w)[R], const float (&tau)[16], cf *lds, int tid) const {
        static_assert(R == 16, "radix-16 first pass");
        constexpr int NS = PL::ns(0);
        const int base = (tid / NS) * (NS * 16) + (tid % NS);
        auto store = [&](int k, cf val) __attribute__((always_inline)) { lds[phys<0>(base + k * NS)] = rot_tan(val, tau[k]); };
        dft16s_es_win(v, w, store);
    }
    // butterflies of the LAST pass P on pre-rotated inputs (registers only)
    template <int P> __device__ __forceinline__ void bfly_prerot(cf (&v)[R], int tid) const {
        (void)tid;
        static_assert(P > 0 && PL::radix(P) == 16 && R == 16, "a twiddled radix-16 pass");
        dft16s<true, true>(v, t16[P - 1]);
    }

    // pass 0 on RAW samples with the window folded into the first radix-4 stage (dft16s_es_win), scatter folded in
    __device__ __forceinline__ void bfly_scatter_win(cf (&v)[R], const float (&w)[R], cf *lds, int tid) const {
        static_assert(R == 16, "radix-16 first pass");
        constexpr int NS = PL::ns(0);
        const int base = (tid / NS) * (NS * 16) + (tid % NS);
        auto store = [&](int k, cf val) __attribute__((always_inline)) { lds[phys<0>(base + k * NS)] = val; };
        dft16s_es_win(v, w, store);
    }

    template <int P, bool SINGLE, bool WL = false> __device__ __forceinline__ void pass(cf (&v)[R], cf *lds0, cf *lds1, int tid) const {
        constexpr bool LAST = (P == NP - 1);
        cf *lds = (P & 1) ? lds1 : lds0;
        if constexpr (SP_EARLY_SCATTER && !SP_PACKED && !SP_ABLATE && !SINGLE && !LAST && PL::radix(P) == 16 && R == 16) {
            bfly_scatter<P>(v, lds, tid);
            xsync<WL>();
            gather<P>(v, lds, tid);
            pass<P + 1, SINGLE, WL>(v, lds0, lds1, tid);
            return;
        }
        if constexpr (!LAST && SINGLE && !(SP_ABLATE & 2)) xsync<WL>();       // previous readers of this image are done
        bfly<P>(v, tid);
        if constexpr (!LAST && (SP_ABLATE & 2)) {
            pass<P + 1, SINGLE, WL>(v, lds0, lds1, tid);
        } else if constexpr (!LAST) {
            scatter<P>(v, lds, tid);
            xsync<WL>();
            gather<P>(v, lds, tid);
            pass<P + 1, SINGLE, WL>(v, lds0, lds1, tid);
        }
    }
};

}   // namespace sp
