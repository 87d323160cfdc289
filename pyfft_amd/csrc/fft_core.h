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

// SP_ABLATE (diagnostic builds only, results wrong): bit 0 = skip the radix-16 butterflies, bit 1 = skip the LDS
// exchanges (and their barriers), bit 2 = skip the inter-pass twiddle multiplies
#ifndef SP_ABLATE
#define SP_ABLATE 0
#endif
template <int RDX, bool INV> __device__ __forceinline__ void dftR(cf (&x)[RDX]) {
    if (RDX == 16 && (SP_ABLATE & 1)) return;
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
    static constexpr int PITCH1 = T + 1;
    static constexpr int LDS_ELEMS = NP > 1 ? (R * PITCH1 > N ? R * PITCH1 : N) : 0;
};

// ---- the workgroup FFT ------------------------------------------------------------------
// TWL: the twiddles of pass 1 (radix 16 after one radix-16 pass: W_256^{(q%16) s}, only 16 x 15 distinct values per
// workgroup) are read from a small LDS table `ltw[(s-1)*16 + q%16]` each frame instead of living in 30 VGPRs.
template <int N, bool INV, bool TWL = false> struct WgFft {
    using PL = FftPlan<N>;
    static constexpr int R = PL::R, T = PL::T, NP = PL::NP;
    static constexpr bool USE_TWL = TWL && NP >= 2 && PL::radix(1) == 16 && PL::radix(0) == 16;
    static constexpr int LTW_ELEMS = USE_TWL ? 15 * 16 : 0;
    cf tw[PL::NTW > 0 ? PL::NTW : 1];
    const cf *ltw = nullptr;

    // table[m] = exp(-2 pi i m / N), m = 0..N-1
    __device__ __forceinline__ void load_twiddles(const cf *__restrict__ table, int tid) { load_tw<1>(table, tid); }

    // fill the LDS table (every thread of the workgroup calls this once, then a barrier)
    __device__ __forceinline__ void fill_lds_twiddles(const cf *__restrict__ table, cf *lds_table, int wg_tid, int wg_size) {
        if constexpr (USE_TWL) {
            for (int e = wg_tid; e < 15 * 16; e += wg_size) {
                const int s = e / 16 + 1, j = e % 16;
                lds_table[e] = table[(j * s) * (N / 256)];
            }
            ltw = lds_table;
        }
    }

    template <int P> __device__ __forceinline__ void load_tw(const cf *__restrict__ table, int tid) {
        if constexpr (P < NP) {
            constexpr int r = PL::radix(P), NS = PL::ns(P), NB = R / r, OFF = PL::ntw_before(P);
            if constexpr (!(USE_TWL && P == 1)) {
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int q = tid + T * u;
                    const int e = (q % NS) * (N / (NS * r));
#pragma unroll
                    for (int s = 1; s < r; ++s) tw[OFF + u * (r - 1) + (s - 1)] = table[e * s];
                }
            }
            load_tw<P + 1>(table, tid);
        }
    }

    // physical LDS index of logical element i for exchange number E (0 = first)
    template <int E> static __device__ __forceinline__ int phys(int i) {
        if constexpr (E == 0) return (i % PL::radix(0)) * PL::PITCH1 + i / PL::radix(0);
        else return i;
    }

    // v[t] <-> element tid + T*t on entry and exit.  lds0/lds1: two exchange images (may be equal,
    // SINGLE=true, then a barrier also precedes every write).
    template <bool SINGLE> __device__ __forceinline__ void run(cf (&v)[R], cf *lds0, cf *lds1, int tid) const {
        pass<0, SINGLE>(v, lds0, lds1, tid);
    }

    template <int P, bool SINGLE> __device__ __forceinline__ void pass(cf (&v)[R], cf *lds0, cf *lds1, int tid) const {
        constexpr int r = PL::radix(P), NS = PL::ns(P), NB = R / r, OFF = PL::ntw_before(P);
        constexpr bool LAST = (P == NP - 1);
        cf *lds = (P & 1) ? lds1 : lds0;
        if constexpr (!LAST && SINGLE && !(SP_ABLATE & 2)) __syncthreads();   // previous readers of this image are done
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            cf x[r];
#pragma unroll
            for (int s = 0; s < r; ++s) x[s] = v[u + s * NB];
            if constexpr (P > 0 && !(SP_ABLATE & 4)) {
                if constexpr (USE_TWL && P == 1) {
                    const int j = (tid + T * u) & 15;
#pragma unroll
                    for (int s = 1; s < r; ++s) x[s] = twm<INV>(x[s], ltw[(s - 1) * 16 + j]);
                } else {
#pragma unroll
                    for (int s = 1; s < r; ++s) x[s] = twm<INV>(x[s], tw[OFF + u * (r - 1) + (s - 1)]);
                }
            }
            dftR<r, INV>(x);
            if constexpr (LAST || (SP_ABLATE & 2)) {
#pragma unroll
                for (int s = 0; s < r; ++s) v[u + s * NB] = x[s];
            } else {
                const int q = tid + T * u;
                const int base = (q / NS) * (NS * r) + (q % NS);
#pragma unroll
                for (int s = 0; s < r; ++s) lds[phys<P>(base + s * NS)] = x[s];
            }
        }
        if constexpr (!LAST && (SP_ABLATE & 2)) {
            pass<P + 1, SINGLE>(v, lds0, lds1, tid);
        } else if constexpr (!LAST) {
            __syncthreads();
            if constexpr (P == 0 && (T % 16) == 0 && PL::radix(0) == 16) {
                // i = tid + T*t  ->  (i%16)*PITCH1 + i/16 = (tid%16)*PITCH1 + tid/16 + (T/16)*t
                const int b = (tid % 16) * PL::PITCH1 + tid / 16;
#pragma unroll
                for (int t = 0; t < R; ++t) v[t] = lds[b + (T / 16) * t];
            } else {
#pragma unroll
                for (int t = 0; t < R; ++t) v[t] = lds[phys<P>(tid + T * t)];
            }
            pass<P + 1, SINGLE>(v, lds0, lds1, tid);
        }
    }
};

}   // namespace sp
