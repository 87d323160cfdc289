// kernels.h -- gfx950 kernels of the spectral hot path, all built on WgFft (fft_core.h).
//
// Layout in HBM: signals are contiguous sample vectors (float32 or interleaved complex64);
// frames are never materialised -- frame g is the window [g*hop, g*hop+n) of the signal.
// A workgroup hosts FPW = 256/T transform groups (T = L/16 threads each); a group owns a run
// of consecutive frames (Welch/STFT) or a strided set of rows/blocks (FFT/Hilbert/FIR).
//
// Every kernel is templated on a transform policy X:
//   XfPow2<N>  : n == N, one workgroup Stockham FFT (the fast path, BASELINE sizes);
//   XfBlue<L>  : any n with 2n-1 <= L: Bluestein chirp-z  X[k] = c*[k] IFFT_L(FFT_L(x c*) . FFT_L(c))[k]
//                with c[m] = exp(i pi m^2/n); both L-point transforms run in the same workgroup,
//                so non-power-of-two segment lengths (the reference's usual case: nwins =
//                floor(nsig/(Navr(1-ov)+ov))) stay one fused kernel.
// Register contract of X::fwd: v[t] <-> element tid + T*t (t < 16), valid for indices < n.
#pragma once
#include "fft_core.h"
#include <stdint.h>

namespace sp {

// SP_CARRY_WAVELOCAL=1: k_welch_carry at <= 1024 points exchanges without workgroup barriers (one wave holds a whole transform)
#ifndef SP_CARRY_WAVELOCAL
#define SP_CARRY_WAVELOCAL 1
#endif
template <int L> struct WgCfg {
    using PL = FftPlan<L>;
    static constexpr int R = PL::R, T = PL::T;
    static constexpr int WG = T >= 256 ? T : 256;
    static constexpr int FPW = WG / T;
    static constexpr int LDS_PER = PL::LDS_ELEMS;              // complex elements per transform image
    static constexpr size_t lds_bytes(int nbuf) { return (size_t)FPW * LDS_PER * sizeof(cf) * nbuf; }
};

// the same with the workgroup widened WM times (more transforms per workgroup, same per-transform constants)
template <int L, int WM> struct WgCfgW {
    using B = WgCfg<L>;
    static constexpr int R = B::R, T = B::T, WG = B::WG * WM, FPW = B::FPW * WM, LDS_PER = B::LDS_PER;
};

// device tables of one transform length
struct XfTables {
    const cf *tw;      // exp(-2 pi i m/L), m < L
    const cf *chirp;   // Bluestein: exp(-i pi m^2/n), m < n          (null for pow2)
    const cf *bf;      // Bluestein: FFT_L(wrapped exp(+i pi m^2/n)) / L (null for pow2)
    int n;             // transform length
};

template <int N, bool TW1LDS = false> struct XfPow2 {
    static constexpr int L = N;
    static constexpr bool EXACT = true;      // n == L at compile time
    using C = WgCfg<N>;
    WgFft<N, TW1LDS> f;
    __device__ __forceinline__ void init(const XfTables &tb, int tid) { f.load_twiddles(tb.tw, tid); }
    __device__ __forceinline__ void fwd(cf (&v)[C::R], cf *lds, int tid, int) const { f.template run<true>(v, lds, lds, tid); }
    // two exchange images (ping-pong): one barrier per exchange instead of two
    __device__ __forceinline__ void fwd2(cf (&v)[C::R], cf *lds_a, cf *lds_b, int tid) const { f.template run<false>(v, lds_a, lds_b, tid); }
    // fwd without workgroup barriers, for row-mapped kernels whose transform sits in one wave (T <= 64; fft_core.h, WL)
    __device__ __forceinline__ void fwdw(cf (&v)[C::R], cf *lds, int tid, int) const { f.template run<true, true>(v, lds, lds, tid); }
};

// forward transform of a ROW-MAPPED kernel (SP_KERNEL_PROLOGUE: thread % T within a group of T consecutive threads): power-of-two
// transforms whose T threads are lanes of one wave exchange without workgroup barriers (SP_ROW_WAVELOCAL; fft_core.h, WL) -- the
// group's image is private to it, and the other groups of the workgroup (other waves) are not held up at an s_barrier
#ifndef SP_ROW_WAVELOCAL
#define SP_ROW_WAVELOCAL 1
#endif
template <class X, int R_> __device__ __forceinline__ void fwd_row(const X &xf, cf (&v)[R_], cf *lds, int tid, int n) {
    if constexpr (X::EXACT && SP_ROW_WAVELOCAL) {
        if constexpr (X::C::T <= 64 && X::C::FPW > 1) {
            xf.fwdw(v, lds, tid, n);
            return;
        }
    }
    xf.fwd(v, lds, tid, n);
}

template <int L_> struct XfBlue {
    static constexpr int L = L_;
    static constexpr bool EXACT = false;
    using C = WgCfg<L_>;
    WgFft<L_> f;
    const cf *chirp, *bf;
    __device__ __forceinline__ void init(const XfTables &tb, int tid) {
        f.load_twiddles(tb.tw, tid);
        chirp = tb.chirp;
        bf = tb.bf;
    }
    __device__ __forceinline__ void fwd(cf (&v)[C::R], cf *lds, int tid, int n) const {
        cf c[C::R];
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int i = tid + C::T * t;
            c[t] = chirp[i < n ? i : 0];
            v[t] = i < n ? cmul(v[t], c[t]) : mk(0.f, 0.f);
        }
        f.template run<true>(v, lds, lds, tid);
#pragma unroll
        for (int t = 0; t < C::R; ++t) v[t] = cconj(cmul(v[t], bf[tid + C::T * t]));     // conj: inverse via forward
        f.template run<true>(v, lds, lds, tid);
#pragma unroll
        for (int t = 0; t < C::R; ++t) v[t] = cmul(cconj(v[t]), c[t]);
    }
};

enum { SIDED_ONE = 1, SIDED_TWO = 2, SIDED_RAW = 3, SIDED_HALF = 4 };   // HALF: bins 0..n/2 (numpy rfft), no doubling

// bin k of an n-point spectrum -> output slot for a sidedness (fft_analysis.py:2179-2193 / :402-428);
// -1 when the bin is dropped.  nny = n/2 (even) or (n+1)/2 (odd)  (fft_analysis.py:2471-2484)
__device__ __forceinline__ int nyq_of(int n) { return (n & 1) ? (n + 1) / 2 : n / 2; }
__device__ __forceinline__ int bin_slot(int k, int n, int sided) {
    if (sided == SIDED_ONE) return k < nyq_of(n) ? k : -1;
    if (sided == SIDED_HALF) return k <= n / 2 ? k : -1;
    if (sided == SIDED_TWO) {
        const int s = k + n / 2;
        return s >= n ? s - n : s;
    }
    return k;
}
// [1:-1] of the cropped array is doubled, plus the last bin when n is odd (Q1; :414-420, :2186-2188)
__device__ __forceinline__ bool bin_doubled(int k, int n, int sided) {
    const int nny = nyq_of(n);
    return sided == SIDED_ONE && k >= 1 && (k <= nny - 2 || ((n & 1) && k == nny - 1));
}
__device__ __forceinline__ int nbins_of(int n, int sided) {
    return sided == SIDED_ONE ? nyq_of(n) : (sided == SIDED_HALF ? n / 2 + 1 : n);
}

__device__ __forceinline__ cf load_sample(const void *x, int64_t i, bool cplx) {
    if (cplx) return reinterpret_cast<const cf *>(x)[i];
    return mk(reinterpret_cast<const float *>(x)[i], 0.f);
}

// read-once input streams: non-temporal loads (SP_NT_STREAM_LOADS=0 restores plain loads)
#ifndef SP_NT_STREAM_LOADS
#define SP_NT_STREAM_LOADS 1
#endif
__device__ __forceinline__ float ld_stream(const float *p) {
#if SP_NT_STREAM_LOADS
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ cf ld_stream(const cf *p) {
#if SP_NT_STREAM_LOADS
    typedef float f2_ __attribute__((ext_vector_type(2)));
    const f2_ r = __builtin_nontemporal_load(reinterpret_cast<const f2_ *>(p));
    return mk(r.x, r.y);
#else
    return *p;
#endif
}
// write-once output streams (spectrograms, filtered signals): non-temporal stores (SP_NT_STORES=0 restores plain stores)
#ifndef SP_NT_STORES
#define SP_NT_STORES 1
#endif
typedef float sp_f2s __attribute__((ext_vector_type(2)));
typedef float sp_f4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_stream(float *p, float v) {
#if SP_NT_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ void st_stream(cf *p, cf v) {
#if SP_NT_STORES
    const sp_f2s q = {v.x, v.y};
    __builtin_nontemporal_store(q, reinterpret_cast<sp_f2s *>(p));
#else
    *p = v;
#endif
}
__device__ __forceinline__ void st_stream(float4 *p, float4 v) {
#if SP_NT_STORES
    const sp_f4s q = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(q, reinterpret_cast<sp_f4s *>(p));
#else
    *p = v;
#endif
}

// sum over the 64 lanes of a wave without the LDS pipe: xor-butterfly inside each row of 16 lanes with DPP (quad_perm
// [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror -- after each step the partner groups hold equal values, so the mirrors
// act as xor 4 / xor 8), then the four row sums through v_readlane.  The result is uniform.
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum64(float v) {
    v = dpp_add<0xB1>(v);
    v = dpp_add<0x4E>(v);
    v = dpp_add<0x141>(v);
    v = dpp_add<0x140>(v);
    const int b = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(b, 32)) + __int_as_float(__builtin_amdgcn_readlane(b, 48)));
}

// sum over the lanes of one frame group inside a wave (W = min(T, 64) lanes, a power of two); every lane gets the sum
template <int W> __device__ __forceinline__ float group_lane_sum(float v) {
    if constexpr (W == 64) {
        return wave_sum64(v);
    } else {
#pragma unroll
        for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
        return v;
    }
}

// detrend parameters of one signal: value removed at global sample index i is  m + s*i
struct Trend {
    cf m, s;
};
// (kept in VGPRs: the record is uniform, hipcc would hold it in SGPRs, and a VALU instruction with an SGPR source issues
//  at half rate on gfx950 -- it is subtracted from every sample of every frame)
__device__ __forceinline__ Trend load_trend(const float *p) {
    Trend t{mk(p[0], p[1]), mk(p[2], p[3])};
    asm volatile("" : "+v"(t.m.x), "+v"(t.m.y), "+v"(t.s.x), "+v"(t.s.y));
    return t;
}
template <bool LIN> __device__ __forceinline__ cf detrended(cf a, const Trend &tr, int64_t i) {
    if constexpr (LIN) {
        const float fi = (float)i;
        return mk(a.x - (tr.m.x + tr.s.x * fi), a.y - (tr.m.y + tr.s.y * fi));
    } else {
        return a - tr.m;
    }
}

#define SP_KERNEL_PROLOGUE(X)                                                                         \
    using C = typename X::C;                                                                          \
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];                          \
    cf *smem = reinterpret_cast<cf *>(smem_raw);                                                      \
    /* T >= 64: a wave never straddles two groups -> the group index is wave-uniform: keep it (and every frame / row    \
       base derived from it) in SGPRs instead of 64-bit VGPR arithmetic */                                            \
    const int grp = C::FPW == 1 ? 0 : ((C::T % 64) == 0 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / C::T) : (int)threadIdx.x / C::T); \
    const int tid = C::FPW == 1 ? (int)threadIdx.x : (int)threadIdx.x % C::T;                        \
    cf *lds = smem + grp * C::LDS_PER;                                                                \
    const int n = X::EXACT ? X::L : tb.n;                                                             \
    X xf;                                                                                             \
    xf.init(tb, tid);

// mean of one frame spread over the T threads of a group (v[t] <-> sample tid + T t, slots >= n excluded): the
// per-segment detrend of the matplotlib.mlab estimators (fft_analysis.py:1060-1155 psd / csd / coh).  Reduction through
// the group's exchange image; every thread of the workgroup must call it (barriers).
template <class C> __device__ __forceinline__ cf group_mean(const cf (&v)[C::R], cf *lds, int tid, int n, bool exact) {
    cf s = mk(0.f, 0.f);
#pragma unroll
    for (int t = 0; t < C::R; ++t) {
        const float keep = (exact || tid + C::T * t < n) ? 1.f : 0.f;
        s = s + keep * v[t];
    }
    if constexpr (C::T == 1) {
        return (1.f / (float)n) * s;
    } else {
        __syncthreads();                      // the image may still be read by the previous transform
        lds[tid] = s;
        __syncthreads();
        constexpr int W = C::T < 16 ? C::T : 16;
        cf p = mk(0.f, 0.f);
        if (tid < W) {
            for (int j = tid; j < C::T; j += W) p = p + lds[j];
        }
        __syncthreads();
        if (tid < W) lds[tid] = p;
        __syncthreads();
        cf tot = mk(0.f, 0.f);
#pragma unroll
        for (int j = 0; j < W; ++j) tot = tot + lds[j];
        return (1.f / (float)n) * tot;
    }
}

// per-segment detrend of one frame in place: mode 1 removes its mean, mode 2 its least-squares line
// (slope = sum (i - ibar) x_i / sum (i - ibar)^2, sum (i - ibar)^2 = n (n^2 - 1) / 12).  Workgroup-uniform calls only.
template <class C> __device__ __forceinline__ void segment_detrend(cf (&v)[C::R], cf *lds, int tid, int n, bool exact, int mode) {
    const cf m = group_mean<C>(v, lds, tid, n, exact);
    if (mode == 2 && n > 1) {
        const float ibar = 0.5f * (float)(n - 1);
        cf u[C::R];
#pragma unroll
        for (int t = 0; t < C::R; ++t) u[t] = ((float)(tid + C::T * t) - ibar) * v[t];
        const cf su = group_mean<C>(u, lds, tid, n, exact);                  // (1/n) sum (i - ibar) x_i
        const float inv = 12.f / ((float)n * (float)n - 1.f);                 // n / sum (i - ibar)^2  (times 1/n above)
        const cf sl = inv * su;
#pragma unroll
        for (int t = 0; t < C::R; ++t) v[t] = v[t] - m - ((float)(tid + C::T * t) - ibar) * sl;
    } else {
#pragma unroll
        for (int t = 0; t < C::R; ++t) v[t] = v[t] - m;
    }
}

// ------------------------------------------------------------------------------------------
// A7  batched C2C FFT (fft_analysis.py:2096-2116).  inverse through conj(fft(conj(.)))/n.
// BigTw (optional): after the transform, element (row b, column i) is multiplied by W_Ntot^{b*i} -- the twiddle
// step of the multi-pass large FFT; W is looked up as hi[m >> lb] * lo[m & (2^lb-1)] (two 8192-entry tables keep
// the phase accurate to float rounding for Ntot up to 2^26).
// ------------------------------------------------------------------------------------------
struct BigTw {
    const cf *hi, *lo;
    int lb;
    int64_t mod;       // > 0: rows are numbered modulo `mod` (a batch of independent long transforms in one launch)
};

#ifndef SP_BIGTW_REC
#define SP_BIGTW_REC 1
#endif
// v[t] *= W^{e0 + es t}, t = 0..15, from TWO table look-ups (W^{e0}, W^{es}): the powers of the step by squaring (s, s^2,
// s^4, s^8) and at most three more products, so every factor is <= 7 complex multiplications away from a table entry
// (phase error <= 5e-7).  The direct form is 16 look-ups of two 8-byte loads each whose addresses differ in every lane.
__device__ __forceinline__ void bigtw_apply16(cf (&v)[16], const BigTw &bt, int64_t e0, int64_t es) {
    auto look = [&](int64_t m) __attribute__((always_inline)) { return cmul(bt.hi[m >> bt.lb], bt.lo[m & ((1 << bt.lb) - 1)]); };
    const cf w0 = look(e0), s1 = look(es);
    const cf s2 = cmul(s1, s1), s4 = cmul(s2, s2), s8 = cmul(s4, s4);
    const cf q3 = cmul(s2, s1), q5 = cmul(s4, s1), q6 = cmul(s4, s2), q9 = cmul(s8, s1), q10 = cmul(s8, s2), q12 = cmul(s8, s4);
    const cf q7 = cmul(q6, s1), q11 = cmul(q10, s1), q13 = cmul(q12, s1), q14 = cmul(q12, s2);
    const cf q15 = cmul(q14, s1);
    const cf q[16] = {mk(1.f, 0.f), s1, s2, q3, s4, q5, q6, q7, s8, q9, q10, q11, q12, q13, q14, q15};
    v[0] = cmul(v[0], w0);
#pragma unroll
    for (int t = 1; t < 16; ++t) v[t] = cmul(v[t], cmul(w0, q[t]));
}

template <class X>
__global__ __launch_bounds__(X::C::WG) void k_fft_c2c(const cf *__restrict__ in, cf *__restrict__ out, int64_t batch,
                                                       int inverse, XfTables tb, BigTw bt) {
    SP_KERNEL_PROLOGUE(X)
    const float sgn = inverse ? -1.f : 1.f;
    const float scl = inverse ? 1.f / (float)n : 1.f;
    const int64_t stride = (int64_t)gridDim.x * C::FPW;
    for (int64_t b0 = (int64_t)blockIdx.x * C::FPW; b0 < batch; b0 += stride) {
        const int64_t b = b0 + grp;
        const bool act = b < batch;
        const int64_t bl = act ? b : batch - 1;          // clamped: loads stay unconditional
        cf v[C::R];
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int i = tid + C::T * t;
            v[t] = ld_stream(in + bl * n + (X::EXACT || i < n ? i : n - 1));
        }
#pragma unroll
        for (int t = 0; t < C::R; ++t) v[t] = mk(v[t].x, sgn * v[t].y);
        fwd_row(xf, v, lds, tid, n);
        if (bt.lo != nullptr) {
            const int64_t rowm = bt.mod > 0 ? bl % bt.mod : bl;
#if SP_BIGTW_REC
            if constexpr (C::R == 16) {
                bigtw_apply16(v, bt, rowm * (int64_t)tid, rowm * (int64_t)C::T);
            } else
#endif
            {
#pragma unroll
                for (int t = 0; t < C::R; ++t) {
                    const int64_t m = rowm * (int64_t)(tid + C::T * t);
                    v[t] = cmul(v[t], cmul(bt.hi[m >> bt.lb], bt.lo[m & ((1 << bt.lb) - 1)]));
                }
            }
        }
        if (act) {
#pragma unroll
            for (int t = 0; t < C::R; ++t) {
                const int i = tid + C::T * t;
                if (X::EXACT || i < n) st_stream(out + b * n + i, mk(scl * v[t].x, sgn * scl * v[t].y));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Strided form for the two-pass long transform (N = N1 N2 > 8192): element i of row b sits at in[b*in_rs + i*in_es] and
// goes to out[b*out_rs + i*out_es], so a "row" can be a COLUMN of a row-major matrix and the three explicit transposes
// of the four-step algorithm disappear (2 passes x 16 B/point instead of 5).  A column read touches one 8-byte element
// per 128-byte line, so the lines must be shared in L2: consecutive workgroups of one XCD (blockIdx = xcd + 8 j: the
// hardware deals workgroups to the 8 XCDs in turn) take ADJACENT columns, which are launched and run close together.
// conj_in / conj_out + scale implement the inverse as conj(fft(conj(.)))/N across the two passes; bt (optional) is the
// inter-pass twiddle W_N^{b i}.  Power-of-two lengths.
// ------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(WgCfg<N>::WG) void k_fft_strided(const cf *__restrict__ in, cf *__restrict__ out, int64_t batch,
                                                               int64_t in_rs, int64_t in_es, int64_t out_rs, int64_t out_es,
                                                               int conj_in, int conj_out, float scale, XfTables tb, BigTw bt,
                                                               int remap) {
    using X = XfPow2<N>;
    SP_KERNEL_PROLOGUE(X)
    (void)n;
    const float si = conj_in ? -1.f : 1.f, so = conj_out ? -1.f : 1.f;
    // blocks per XCD = gridDim.x / 8 (the launcher makes the grid a multiple of 8): XCD x owns a contiguous range of rows
    const int64_t per_xcd = gridDim.x / 8;
    const int64_t gid = remap ? (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3) : (int64_t)blockIdx.x;
    const int64_t b = gid * C::FPW + grp;
    if (b >= batch) return;                       // whole groups only (batch is a multiple of FPW); no barrier is skipped
    cf v[C::R];                                   // by a partial workgroup because FPW divides the grid's row count
#pragma unroll
    for (int t = 0; t < C::R; ++t) v[t] = in[b * in_rs + (int64_t)(tid + C::T * t) * in_es];
#pragma unroll
    for (int t = 0; t < C::R; ++t) v[t] = mk(v[t].x, si * v[t].y);
    xf.fwd(v, lds, tid, N);
    if (bt.lo != nullptr) {
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int64_t m = b * (int64_t)(tid + C::T * t);
            v[t] = cmul(v[t], cmul(bt.hi[m >> bt.lb], bt.lo[m & ((1 << bt.lb) - 1)]));
        }
    }
#pragma unroll
    for (int t = 0; t < C::R; ++t) out[b * out_rs + (int64_t)(tid + C::T * t) * out_es] = mk(scale * v[t].x, so * scale * v[t].y);
}

// ------------------------------------------------------------------------------------------
// Three-pass long transform, N = A B C (n = a BC + b C + c,  k = ka + A kb + A B kc), every pass reading and writing
// whole 128-byte lines and no transposes (3 x 16 B/point against the five passes of the four-step form):
//   1  for every column m = b C + c:   T[ka][m]      = W_N^{m ka}     sum_a x[a][m]     W_A^{a ka}      (k_fft_cols)
//   2  for every ka and column c:      T[ka][kb][c]  = W_N^{A c kb}   sum_b T[ka][b][c] W_B^{b kb}      (k_fft_cols, in place)
//   3  for every row (ka, kb):         X[ka + A kb + A B kc] = sum_c T[ka][kb][c] W_C^{c kc}            (k_fft_rows_rev)
// k_fft_cols: a workgroup owns FPW ADJACENT columns and its lanes run over the columns (group = thread % FPW, the
// flipped mapping), so that element i of the FPW columns is one contiguous segment (128 B at L = 256).
// ------------------------------------------------------------------------------------------
// Fused input forms of the FIRST pass (the element index base + i*es is then the sample / bin index of the whole
// transform), so that the elementwise kernels in front of a long transform and their round trips through HBM disappear:
//   kind 1: real samples, zero-padded: (r1[i] - m1, r2 ? r2[i] - m2 : 0) for i < nreal, else 0  (Hilbert's real -> complex
//           pack; ccf's z = (x1 - m1) + i (x2 - m2): mom[0], mom[1] are the means; the zero half is not even loaded)
//   kind 3: real samples taken in PAIRS, zero-padded: (r1[2i], r1[2i+1]) -- the half-length transform of a real signal
//           (long Hilbert: z[n] = x[2n] + i x[2n+1], M = N/2 points)
// (ccf's middle step, R[k] from Z[k] and Z[L-k], was tried as a third form: 388 VGPRs, 132 spilled at the 2-wave cap --
//  it stays its own kernel, k_xc_mid; so does the half-length Hilbert's, k_hilbert_mid)
// workgroup index -> block index with blocks 2m, 2m + 1 on workgroups w, w + 8 (a bijection on every aligned run of 16)
__device__ __forceinline__ int64_t xcd_pair(int64_t w) { return (w & ~(int64_t)15) | ((w & 7) << 1) | ((w >> 3) & 1); }
struct ColsIn {
    int kind;
    const float *r1, *r2;
    const double *mom;
    int64_t nreal;
};
// HM: the analytic-signal mask of the full-length Hilbert inverse (hmask_n > 0) as a template flag too -- its sixteen 64-bit bin
// indices cost 65 VGPRs; without it the plain form (KIND 0) fits 3 waves per SIMD (166 VGPRs), i.e. three workgroups per CU
// instead of two to cover the load round trips of this unpipelined loop (SP_COLS_WAVES)
#ifndef SP_COLS_WAVES
#define SP_COLS_WAVES 3
#endif
// SP_COLS_PREFETCH=1: the plain form (KIND 0, no mask) loads the NEXT block's column elements before it transforms this one, at two
// workgroups per CU (32 more registers) instead of three unpipelined ones
#ifndef SP_COLS_PREFETCH
#define SP_COLS_PREFETCH 0
#endif
// WM: workgroup width multiplier -- WM = 2 gives a workgroup 2 FPW adjacent columns (512 threads at L = 256: 32 columns, so every
// row access is 256 contiguous bytes of complex data / 128 of real samples instead of 128 / 64).  Used for the FIRST pass of the long
// transforms (KIND 1 / 3: real samples at a stride of B C -- 64-byte pieces a megabyte apart ran at 2.0-2.7 TB/s)
template <int L, int KIND, bool HM = false, int WM = 1>   // KIND = ci.kind as a template parameter: as a run-time branch the load forms cost 306 VGPRs
__global__ __launch_bounds__(WgCfg<L>::WG * WM)
    __attribute__((amdgpu_waves_per_eu((((KIND == 0 && !SP_COLS_PREFETCH) || KIND == 4) && !HM && L <= 256) ? SP_COLS_WAVES : 2,
                                       (((KIND == 0 && !SP_COLS_PREFETCH) || KIND == 4) && !HM && L <= 256) ? SP_COLS_WAVES : 2))) void k_fft_cols(const cf *__restrict__ in, cf *__restrict__ out, int64_t ncolblocks,
                                                            int64_t nouter, int64_t es, int64_t os, int64_t twmul, int conj_in,
                                                            XfTables tb, BigTw bt, int64_t hmask_n, ColsIn ci, int tw_outer = 0) {
    using X = XfPow2<L>;
    using C = WgCfgW<L, WM>;                      // the plan's constants with the widened workgroup
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *smem = reinterpret_cast<cf *>(smem_raw);
    const int grp = C::FPW == 1 ? 0 : (int)threadIdx.x % C::FPW;
    const int tid = C::FPW == 1 ? (int)threadIdx.x : (int)threadIdx.x / C::FPW;
    cf *lds = smem + grp * C::LDS_PER;
    X xf;
    xf.init(tb, tid);
    const float si = conj_in ? -1.f : 1.f;
    // a workgroup walks over (outer, column block) pairs with the grid as stride: the twiddle set-up is paid once per
    // workgroup, and at any time neighbouring workgroups read neighbouring 128-byte segments of the same rows
    const int64_t total = nouter * ncolblocks;
    constexpr bool PF = SP_COLS_PREFETCH && KIND == 0 && !HM;
    auto base_of = [&](int64_t w) __attribute__((always_inline)) {
        const int64_t ix = (tw_outer & 2) ? xcd_pair(w) : w;
        return (ix / ncolblocks) * os + (ix % ncolblocks) * C::FPW + grp;
    };
    cf vpf[PF ? C::R : 1];
    if constexpr (PF) {
        const int64_t b0 = base_of((int64_t)blockIdx.x < total ? (int64_t)blockIdx.x : 0);
#pragma unroll
        for (int t = 0; t < C::R; ++t) vpf[t] = in[b0 + (int64_t)(tid + C::T * t) * es];
    }
    for (int64_t widx = blockIdx.x; widx < total; widx += gridDim.x) {
        // (tw_outer bit 1, set by the launcher when the grid and the block count are multiples of 16: column blocks 2m and 2m + 1 go
        //  to workgroups 8 apart = the same XCD, so that the two 64-byte halves of a 128-byte line of REAL samples meet in one L2)
        const int64_t idx = (tw_outer & 2) ? xcd_pair(widx) : widx;
        const int64_t col = (idx % ncolblocks) * C::FPW + grp;
        const int64_t base = (idx / ncolblocks) * os + col;
        cf v[C::R];
        if constexpr (KIND == 1) {
            const float m1 = ci.mom ? (float)ci.mom[0] : 0.f, m2 = ci.mom ? (float)ci.mom[1] : 0.f;
#pragma unroll
            for (int t = 0; t < C::R; ++t) {
                const int64_t i = base + (int64_t)(tid + C::T * t) * es;
                const bool ok = i < ci.nreal;
                const int64_t ic = ok ? i : 0;
                const float a = ci.r1[ic], b = ci.r2 ? ci.r2[ic] : m2;
                v[t] = ok ? mk(a - m1, b - m2) : mk(0.f, 0.f);
            }
        } else if constexpr (KIND == 4) {
            // kind 1 when the samples end exactly at the middle row (nreal = L/2 * es: a power-of-two ccf): slots t < R/2 are
            // all samples, the others all padding -- no predicates, no 64-bit compares, and the zeros fold into the first butterfly
            const float m1 = (float)ci.mom[0], m2 = (float)ci.mom[1];
            const int ib = (int)base, ies = (int)es;                 // (32-bit sample indices: at most 2^26 samples per signal)
#pragma unroll
            for (int t = 0; t < C::R; ++t) {
                if (t < C::R / 2) {
                    const int i = ib + (tid + C::T * t) * ies;
                    v[t] = mk(ci.r1[i] - m1, ci.r2[i] - m2);
                } else {
                    v[t] = mk(0.f, 0.f);
                }
            }
        } else if constexpr (KIND == 3) {
            if (ci.r2 != nullptr) {
                // r2 == r1 flags an 8-byte aligned row: one 8-byte load per pair (the scalar form below ran this pass at
                // 1.5 TB/s against 2.7 for the plain complex load)
#pragma unroll
                for (int t = 0; t < C::R; ++t) {
                    const int64_t i = 2 * (base + (int64_t)(tid + C::T * t) * es);
                    const bool ok0 = i < ci.nreal, ok1 = i + 1 < ci.nreal;
                    const cf pr = *reinterpret_cast<const cf *>(ci.r1 + (ok1 ? i : 0));
                    const float a0 = ok1 ? pr.x : (ok0 ? ci.r1[i] : 0.f);
                    v[t] = mk(a0, ok1 ? pr.y : 0.f);
                }
            } else {
#pragma unroll
                for (int t = 0; t < C::R; ++t) {
                    const int64_t i = 2 * (base + (int64_t)(tid + C::T * t) * es);
                    const bool ok0 = i < ci.nreal, ok1 = i + 1 < ci.nreal;
                    const float a = ci.r1[ok0 ? i : 0], b = ci.r1[ok1 ? i + 1 : 0];
                    v[t] = mk(ok0 ? a : 0.f, ok1 ? b : 0.f);
                }
            }
        } else if constexpr (PF) {
            // (no other workgroup writes the next block's columns, also in place: the early read is safe)
            const int64_t bn = base_of(widx + gridDim.x < total ? widx + gridDim.x : widx);
#pragma unroll
            for (int t = 0; t < C::R; ++t) {
                v[t] = vpf[t];
                vpf[t] = in[bn + (int64_t)(tid + C::T * t) * es];
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
            // (32-bit element offsets from the block's base: a column spans less than 2^27 elements)
            const cf *pin = in + base;
            const int ies = (int)es;
#pragma unroll
            for (int t = 0; t < C::R; ++t) v[t] = pin[(tid + C::T * t) * ies];
        }
        if (HM && hmask_n > 0) {
            // analytic-signal mask (hilbert.py:63-64) applied while loading the spectrum for the inverse transform
            // (pass 1 only: base + i*es is the bin index): k = 0 and k = nyq x1, 1..nyq-1 x2, above x0
            const int64_t nyq = (hmask_n & 1) ? (hmask_n + 1) / 2 : hmask_n / 2;
#pragma unroll
            for (int t = 0; t < C::R; ++t) {
                const int64_t k = base + (int64_t)(tid + C::T * t) * es;
                const float h = (k == 0 || k == nyq) ? 1.f : (k < nyq ? 2.f : 0.f);
                v[t] = h * v[t];
            }
        }
#pragma unroll
        for (int t = 0; t < C::R; ++t) v[t] = mk(v[t].x, si * v[t].y);
        xf.fwd(v, lds, tid, L);
        // (tw_outer: the twiddle is indexed by the OUTER index instead of the column -- the ccf's half rows [ka][kb][ka'], where
        //  the M-point transform's column index c' = ka is the outer one of the pass over kb)
        const int64_t mc = twmul * ((tw_outer & 1) ? idx / ncolblocks : col);
#if SP_BIGTW_REC
        if constexpr (C::R == 16) {
            // W^{mc (tid + T t)} = W^{mc tid} (W^{mc T})^t: two table look-ups per thread instead of sixteen (each look-up is
            // two 8-byte loads whose addresses differ in every lane -- 32 scattered loads per thread and block), the powers of
            // the step by squaring (s, s^2, s^4, s^8) and at most three more products: every factor is <= 7 complex
            // multiplications away from a table entry (phase error <= 5e-7)
            bigtw_apply16(v, bt, mc * (int64_t)tid, mc * (int64_t)C::T);
        } else
#endif
        {
#pragma unroll
            for (int t = 0; t < C::R; ++t) {
                const int64_t m = mc * (int64_t)(tid + C::T * t);
                v[t] = cmul(v[t], cmul(bt.hi[m >> bt.lb], bt.lo[m & ((1 << bt.lb) - 1)]));
            }
        }
        {
            cf *pout = out + base;
            const int ies = (int)es;
#pragma unroll
            for (int t = 0; t < C::R; ++t) pout[(tid + C::T * t) * ies] = v[t];
        }
    }
}

// pass 3: rows (ka, kb) of T[ka][kb][c]; the FPW rows of a workgroup are adjacent ka of one kb, so that for one kc the
// workgroup's outputs X[ka + A kb + A B kc] are contiguous
// RowsOut (ccf): instead of the complex spectrum, write co[lag + n - 1] = mom[2] / Ltot * Re X[j] for the lags |lag| < n
// (lag = j for j < n, j - Ltot for j > Ltot - n; see k_xc_out) -- the last elementwise kernel of the long ccf, fused
// RowsOut kind 2 (half-length Hilbert, hilbert.py:54-67): the transform's output is z'[j] = y[2j] + i y[2j+1] with y the
// Hilbert transform of the real signal rx; written is the analytic signal a[2j] = rx[2j] + i Re z'[j], a[2j+1] = rx[2j+1] +
// i Im z'[j] (rx zero-padded beyond n samples) into co seen as complex -- two adjacent complex values per element
struct RowsOut {
    float *co;
    int64_t n, Ltot;
    const double *mom;
    const float *rx = nullptr;
    int kind = 0;            // 0: plain spectrum, 1: ccf lags, 2: analytic signal from the half-length transform,
                             // 3: ccf lags from the half-length inverse (two lags per element)
};
template <int L, int OKIND>
__global__ __launch_bounds__(WgCfg<L>::WG) void k_fft_rows_rev(const cf *__restrict__ in, cf *__restrict__ out, int64_t A, int64_t B,
                                                                int conj_out, float scale, XfTables tb, RowsOut ro) {
    using X = XfPow2<L>;
    SP_KERNEL_PROLOGUE(X)
    (void)n;
    const float so = conj_out ? -1.f : 1.f;
    const int64_t AB = A * B, nblocks = AB / C::FPW;              // FPW divides A
    for (int64_t idx = blockIdx.x; idx < nblocks; idx += gridDim.x) {
        const int64_t j = idx * C::FPW + grp;
        const int64_t ka = j % A, kb = j / A;
        const cf *row = in + (ka * B + kb) * (int64_t)L;
        cf v[C::R];
#pragma unroll
        for (int t = 0; t < C::R; ++t) v[t] = row[tid + C::T * t];
        xf.fwd(v, lds, tid, L);
        const int64_t off = kb * A + ka;
        if constexpr (OKIND == 2) {
            float4 *ao = reinterpret_cast<float4 *>(ro.co);
#pragma unroll
            for (int t = 0; t < C::R; ++t) {
                const int64_t j = (int64_t)(tid + C::T * t) * AB + off;
                const bool ok0 = 2 * j < ro.n, ok1 = 2 * j + 1 < ro.n;
                const float x0 = ro.rx[ok0 ? 2 * j : 0], x1 = ro.rx[ok1 ? 2 * j + 1 : 0];
                ao[j] = make_float4(ok0 ? x0 : 0.f, scale * v[t].x, ok1 ? x1 : 0.f, so * scale * v[t].y);
            }
        } else if constexpr (OKIND == 3) {
            // ccf, half-length inverse: element j holds M (r[2j] - i r[2j+1]); Ltot = 2M is the padded correlation length
            const float nrm = (float)(2.0 * ro.mom[2] / (double)ro.Ltot);
#pragma unroll
            for (int t = 0; t < C::R; ++t) {
                const int64_t j2 = 2 * ((int64_t)(tid + C::T * t) * AB + off);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int64_t idx = j2 + e;
                    const float val = nrm * (e ? -v[t].y : v[t].x);
                    if (idx < ro.n) ro.co[idx + ro.n - 1] = val;
                    else if (idx > ro.Ltot - ro.n) ro.co[idx - ro.Ltot + ro.n - 1] = val;
                }
            }
        } else if constexpr (OKIND == 1) {
            const float nrm = (float)(ro.mom[2] / (double)ro.Ltot);
#pragma unroll
            for (int t = 0; t < C::R; ++t) {
                const int64_t j = (int64_t)(tid + C::T * t) * AB + off;
                if (j < ro.n) ro.co[j + ro.n - 1] = nrm * v[t].x;
                else if (j > ro.Ltot - ro.n) ro.co[j - ro.Ltot + ro.n - 1] = nrm * v[t].x;
            }
        } else {
#pragma unroll
            for (int t = 0; t < C::R; ++t) out[(int64_t)(tid + C::T * t) * AB + off] = mk(scale * v[t].x, so * scale * v[t].y);
        }
    }
}

// ---- half-length Hilbert of a long real row with the middle step inside a ROW pass (round 3) -----------------------------------
// The forward three-pass transform ends with contiguous rows (ka, kb) of C points, and the transform can be inverted by the
// adjoint passes in REVERSED order, which starts with the same rows.  The middle step couples bin k with M - k: for k = ka + A
// (kb + B kc) the partner is row (A - ka, B - 1 - kb), element C - 1 - kc (ka != 0); row (0, B - kb), element C - 1 - kc (ka = 0,
// kb != 0); row (0, 0) itself, element C - kc.  So a workgroup that owns a row AND its mirror row can run
//   forward row FFT -> k_hilbert_mid's arithmetic on (k, M - k) through one LDS exchange -> inverse row FFT
// in place: the forward's last pass store, the middle step's load + store and the inverse's first pass load never touch HBM
// (4 x 64 MB of the 960 MB a 2^24-sample Hilbert moved; 7 launches -> 5).  k_fft_cols_inv are the two adjoint column passes:
// the inter-pass twiddle (conjugate) BEFORE the transform; the last one writes the analytic signal.
template <int L>
__global__ __launch_bounds__(WgCfg<L>::WG) void k_hilbert_rowsmid(cf *__restrict__ Tm, int64_t A, int64_t B, XfTables tb, BigTw btN,
                                                               const cf *__restrict__ tw2c /* exp(-2 pi i m / (2 C)), m < 2 C */) {
    using X = XfPow2<L>;
    SP_KERNEL_PROLOGUE(X)
    (void)n;
    static_assert(C::FPW >= 2 && (C::FPW % 2) == 0, "a workgroup holds rows and their mirror rows");
    constexpr int HP = C::FPW / 2;                               // row pairs per workgroup and iteration
    const int64_t AB = A * B, nslots = AB / 2;                   // pairs + ONE slot for the two self-mirrored rows (0, 0), (0, B/2):
                                                                 // a power of two, so that the workgroups' iteration counts are equal
                                                                 // (with a slot each, one workgroup ran a third iteration after all others' two)
    const int side = grp >= HP ? 1 : 0;
    const int pg = side ? grp - HP : grp;
    auto look = [&](int64_t m) __attribute__((always_inline)) { return cmul(btN.hi[m >> btN.lb], btN.lo[m & ((1 << btN.lb) - 1)]); };
    // (slot arithmetic in 32 bits with shifts: A, B are powers of two and A B <= 2^16 -- the 64-bit divisions cost registers and time)
    const int Ai = (int)A, Bi = (int)B, lbB = __ffs(Bi) - 1, ABh = (int)(AB / 2), nsl = (int)nslots;
    for (int s0 = (int)blockIdx.x * HP; s0 < nsl; s0 += (int)gridDim.x * HP) {
        const int s = s0 + pg;
        const bool slot_ok = s < nsl;
        const int sc = slot_ok ? s : 0;
        int ka, kb;
        bool self = false;
        if (sc < ABh - Bi) {                                     // rows with 1 <= ka < A/2 (mirror: A/2 < ka' <= A - 1)
            const int j = Bi + sc;
            ka = j >> lbB;
            kb = j & (Bi - 1);
        } else {
            const int s2 = sc - (ABh - Bi);
            if (s2 < Bi / 2 - 1) {                               // (0, kb), 1 <= kb < B/2  <->  (0, B - kb)
                ka = 0;
                kb = 1 + s2;
            } else if (s2 < Bi - 1) {                            // (A/2, kb), kb < B/2  <->  (A/2, B - 1 - kb)
                ka = Ai / 2;
                kb = s2 - (Bi / 2 - 1);
            } else {                                             // the two rows that are their own mirror: side 0 takes (0, 0),
                ka = 0;                                          // side 1 takes (0, B/2); each is its own partner
                kb = 0;
                self = true;
            }
        }
        const int kam = ka != 0 ? Ai - ka : 0, kbm = self ? Bi / 2 : (ka != 0 ? Bi - 1 - kb : ((Bi - kb) & (Bi - 1)));
        const int64_t myka = side ? kam : ka, mykb = side ? kbm : kb;
        const bool act = slot_ok;
        cf *row = Tm + (myka * B + mykb) * (int64_t)L;
        cf v[C::R];
#pragma unroll
        for (int t = 0; t < C::R; ++t) v[t] = row[tid + C::T * t];
        xf.fwd(v, lds, tid, L);
        __syncthreads();                                         // every group is done with its transform image
#pragma unroll
        for (int t = 0; t < C::R; ++t) lds[tid + C::T * t] = v[t];
        __syncthreads();
        const cf *pl = smem + (self ? grp : (side ? grp - HP : grp + HP)) * C::LDS_PER;
        const bool zero_row = myka == 0 && mykb == 0;
        cf b[C::R];
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int kc = tid + C::T * t;
            b[t] = pl[zero_row ? ((L - kc) & (L - 1)) : (L - 1 - kc)];
        }
        __syncthreads();                                         // the images are free for the next transform
        // W_N^k = W_N^{ka + A kb} W_{2C}^{kc} (N = 2 A B C): one two-level look-up per row, the rest from the (2 C)-point twiddle
        // table (L1-resident) -- instead of sixteen scattered two-level look-ups per thread
        const cf w0 = look(myka + A * mykb);
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int64_t k = myka + A * (mykb + B * (int64_t)(tid + C::T * t));
            const cf w = cmul(w0, tw2c[tid + C::T * t]);         // W_N^k, N = 2 M
            const cf a = v[t], bb = b[t];
            const cf p = mk(a.x + bb.x, a.y - bb.y), q = mk(a.x - bb.x, a.y + bb.y);          // a + conj b, a - conj b
            const cf r = cmul(cconj(w), p) - cmul(w, q);
            v[t] = k == 0 ? mk(0.f, 0.f) : mk(0.5f * r.x, -0.5f * r.y);                       // conj(Z'[k]): inverse = conj(fft(conj))
        }
        xf.fwd(v, lds, tid, L);
        if (act) {
#pragma unroll
            for (int t = 0; t < C::R; ++t) row[tid + C::T * t] = mk(v[t].x, -v[t].y);
        }
        __syncthreads();
    }
}

// ---- long ccf: the forward transform's last (row) pass, k_xc_mid_half and the FIRST pass of the half-length transform in one kernel
// Z = FFT_L(x1 + i x2) after its two column passes lies as rows (ka, kb) of C points.  The half-length spectrum Zp[k], k < M = L/2,
// needs Z[k], Z[M + k] (same row: kc + C/2), Z[L - k] and Z[M - k] (the mirror row of k_hilbert_rowsmid, elements C - 1 - kc and
// C/2 - 1 - kc): a workgroup that owns a row and its mirror row forms the C/2 values of Zp that belong to each.  And those values,
// k = ka + A kb + A B kc, are exactly the input of the M-point transform under the split (A' = C/2, B' = B, C' = A) with a' = kc:
// its first pass is the (C/2)-point transform along the row it already holds.  So: FFT_C -> middle step (one LDS exchange) ->
// FFT_{C/2} -> the inter-pass twiddle W_M^{(kb A + ka) ka'} -> half rows [ka][kb][ka'] in place (row pitch C); two column passes follow (over
// kb with the twiddle indexed by the OUTER index ka, then over ka), the last one writing the lags.  1.9 GB instead of 3.0 GB at
// 2^24 samples, 5 launches instead of 7.
#ifndef SP_XCROWS_PREFETCH
#define SP_XCROWS_PREFETCH 1
#endif
template <int L>
__global__ __launch_bounds__(WgCfg<L>::WG) void k_xc_rowsmid(cf *__restrict__ Tm, int64_t A, int64_t B, XfTables tb, XfTables tb2,
                                                            BigTw btL, BigTw btM) {
    using X = XfPow2<L>;
    SP_KERNEL_PROLOGUE(X)
    (void)n;
    static_assert(C::FPW >= 2 && (C::FPW % 2) == 0 && L >= 64, "a workgroup holds rows and their mirror rows");
    constexpr int HP = C::FPW / 2, L2 = L / 2, T2 = C::T / 2;
    using F2 = WgFft<L2>;
    static_assert(F2::T == T2 && F2::R == C::R, "the half-length transform takes half the row's threads");
    F2 f2;
    const int tid2 = tid % T2, half = tid / T2;
    f2.load_twiddles(tb2.tw, tid2);
    // a group's LDS region here: L + 32 elements (the launcher sizes the allocation): the C-point image (L + 16) or two (C/2)-point
    // images of L/2 + 16 each
    constexpr int PER = L + 32;
    static_assert(FftPlan<L>::LDS_ELEMS <= PER && 2 * FftPlan<L2>::LDS_ELEMS <= PER, "group region");
    lds = smem + grp * PER;
    cf *lds2 = lds + half * (PER / 2);
    const int64_t AB = A * B, nslots = AB / 2;                   // (one slot for the two self-mirrored rows, see k_hilbert_rowsmid)
    const int side = grp >= HP ? 1 : 0;
    const int pg = side ? grp - HP : grp;
    auto look = [&](const BigTw &bt, int64_t m) __attribute__((always_inline)) { return cmul(bt.hi[m >> bt.lb], bt.lo[m & ((1 << bt.lb) - 1)]); };
    // slot -> the group's row (ka, kb) of the pair (side 0) or its mirror row (side 1); self: the two rows that are their own mirror
    struct Slot {
        int64_t myka, mykb;
        bool act, self;
    };
    const int Ai = (int)A, Bi = (int)B, lbB = __ffs(Bi) - 1, ABh = (int)(AB / 2);        // (32-bit slot arithmetic with shifts, as above)
    auto decode = [&](int64_t s0) __attribute__((always_inline)) {
        const int s = (int)s0 + pg;
        const bool slot_ok = s < (int)nslots;
        const int sc = slot_ok ? s : 0;
        int ka, kb;
        bool self = false;
        if (sc < ABh - Bi) {
            const int j = Bi + sc;
            ka = j >> lbB;
            kb = j & (Bi - 1);
        } else {
            const int s2 = sc - (ABh - Bi);
            if (s2 < Bi / 2 - 1) {
                ka = 0;
                kb = 1 + s2;
            } else if (s2 < Bi - 1) {
                ka = Ai / 2;
                kb = s2 - (Bi / 2 - 1);
            } else {
                ka = 0;
                kb = 0;
                self = true;
            }
        }
        const int kam = ka != 0 ? Ai - ka : 0, kbm = self ? Bi / 2 : (ka != 0 ? Bi - 1 - kb : ((Bi - kb) & (Bi - 1)));
        return Slot{(int64_t)(side ? kam : ka), (int64_t)(side ? kbm : kb), slot_ok, self};
    };
    // the NEXT slot's rows are loaded while this one is transformed (SP_XCROWS_PREFETCH: the loop ran load -> two transforms -> store
    // with two workgroups per CU; no other workgroup writes a slot's rows, so the early read is safe)
    const int64_t sstep = (int64_t)gridDim.x * HP;
    cf v[C::R];
#if SP_XCROWS_PREFETCH
    {
        const Slot s1 = decode((int64_t)blockIdx.x * HP);
        const cf *r1 = Tm + (s1.myka * B + s1.mykb) * (int64_t)L;
#pragma unroll
        for (int t = 0; t < C::R; ++t) v[t] = r1[tid + C::T * t];
    }
#endif
    for (int64_t s0 = (int64_t)blockIdx.x * HP; s0 < nslots; s0 += sstep) {
        const Slot sl = decode(s0);
        const int64_t myka = sl.myka, mykb = sl.mykb;
        const bool act = sl.act, self = sl.self;
        cf *row = Tm + (myka * B + mykb) * (int64_t)L;
#if SP_XCROWS_PREFETCH
        cf vn[C::R];
        {
            const Slot s2 = decode(s0 + sstep < nslots ? s0 + sstep : s0);       // (the last iteration re-reads its own rows: unused)
            const cf *r2 = Tm + (s2.myka * B + s2.mykb) * (int64_t)L;
#pragma unroll
            for (int t = 0; t < C::R; ++t) vn[t] = r2[tid + C::T * t];
        }
        __builtin_amdgcn_sched_barrier(0);
#else
#pragma unroll
        for (int t = 0; t < C::R; ++t) v[t] = row[tid + C::T * t];
#endif
        xf.fwd(v, lds, tid, L);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < C::R; ++t) lds[tid + C::T * t] = v[t];
        __syncthreads();
        const cf *pl = smem + (self ? grp : (side ? grp - HP : grp + HP)) * PER;
        const bool zero_row = myka == 0 && mykb == 0;
        // W_L^k = W_L^{ka + A kb} W_C^{kc} (L = A B C): one look-up in the two-level table per row, the rest from the C-point
        // transform's own twiddle table (4 KiB, L1-resident) -- instead of eight scattered two-level look-ups per thread
        const cf w0 = look(btL, myka + A * mykb);
        cf zp[C::R / 2];
#pragma unroll
        for (int t = 0; t < C::R / 2; ++t) {
            const int kc = tid + C::T * t;                       // < C/2
            const cf a = v[t], bm = v[t + C::R / 2];             // Z[k], Z[M + k]
            const cf am = pl[zero_row ? ((L - kc) & (L - 1)) : (L - 1 - kc)];              // Z[L - k]
            const cf b = pl[zero_row ? ((L2 - kc) & (L - 1)) : (L2 - 1 - kc)];             // Z[M - k]
            const int64_t k = myka + A * (mykb + B * (int64_t)kc);
            const cf za = cmul(a, am), zb = cmul(b, bm);
            const cf rk = mk(0.5f * za.y, 0.25f * (cnorm(a) - cnorm(am)));
            const cf rmc = mk(0.5f * zb.y, -0.25f * (cnorm(b) - cnorm(bm)));              // conj R(M - k)
            (void)k;
            const cf w = cmul(w0, tb.tw[kc]);                                             // W_L^k
            const cf sm = rk + rmc, d = rk - rmc;
            const cf tt = cmul(cconj(w), d);
            zp[t] = mk(0.5f * (sm.x - tt.y), -0.5f * (sm.y + tt.x));
        }
        __syncthreads();                                         // every partner read is done: the images may be overwritten
        // the row's C/2 values, re-dealt to the layout of a (C/2)-point transform (both halves of the group's threads take the
        // same elements and run the same transform on their own image: one barrier structure for the whole workgroup)
#pragma unroll
        for (int t = 0; t < C::R / 2; ++t) lds[tid + C::T * t] = zp[t];
        __syncthreads();
        cf u[C::R];
#pragma unroll
        for (int t = 0; t < C::R; ++t) u[t] = lds[tid2 + T2 * t];
        __syncthreads();
        f2.template run<true>(u, lds2, lds2, tid2);
        // inter-pass twiddle of the M-point transform: W_M^{m ka'}, m = kb A + ka (its column index), ka' = tid2 + T2 t
        {
            const int64_t m = mykb * A + myka;
            if constexpr (C::R == 16) {
                bigtw_apply16(u, btM, m * (int64_t)tid2, m * (int64_t)T2);
            } else {
#pragma unroll
                for (int t = 0; t < C::R; ++t) u[t] = cmul(u[t], look(btM, m * (int64_t)(tid2 + T2 * t)));
            }
        }
        if (act && half == 0) {
            cf *orow = row;                                      // the first C/2 elements of its OWN row: [ka][kb][ka'] at row pitch C
            //                                                      (packed half rows would land on rows other workgroups still read)
#pragma unroll
            for (int t = 0; t < C::R; ++t) orow[tid2 + T2 * t] = u[t];
        }
        __syncthreads();
#if SP_XCROWS_PREFETCH
#pragma unroll
        for (int t = 0; t < C::R; ++t) v[t] = vn[t];
#endif
    }
}

// last column pass of the ccf's half-length transform: plain forward transform along the column (no twiddle).  Columns (outer, col):
// element i of the column at in[outer os + col + i es]; its NATURAL index is j = col + ncols (outer + nouter i) (the half rows lie at
// row pitch C, ncols = C/2 of them used); written are the lags (RowsOut kind 3's output: element j holds M (r[2j] - i r[2j+1]))
#ifndef SP_COLSX_WAVES
#define SP_COLSX_WAVES 3          // k_fft_cols_inv, complex output: three workgroups per CU for this unpipelined loop (161 VGPRs, no spills;
                                  // the forms with the analytic-signal / lag output spill 21-51 registers at three and stay at two)
#endif
template <int L>
__global__ __launch_bounds__(WgCfg<L>::WG) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_fft_cols_lag(
    const cf *__restrict__ in, int64_t ncolblocks, int64_t nouter, int64_t es, int64_t os, XfTables tb, RowsOut ro) {
    using X = XfPow2<L>;
    using C = typename X::C;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *smem = reinterpret_cast<cf *>(smem_raw);
    const int grp = C::FPW == 1 ? 0 : (int)threadIdx.x % C::FPW;
    const int tid = C::FPW == 1 ? (int)threadIdx.x : (int)threadIdx.x / C::FPW;
    cf *lds = smem + grp * C::LDS_PER;
    X xf;
    xf.init(tb, tid);
    const float nrm = (float)(2.0 * ro.mom[2] / (double)ro.Ltot);
    const int64_t ncols = ncolblocks * C::FPW, total = ncolblocks * nouter;
    for (int64_t widx = blockIdx.x; widx < total; widx += gridDim.x) {
        // (ro.kind bit 4: neighbouring column blocks on one XCD -- their lag segments are 128 bytes at an odd float offset, so every
        //  output line is shared by two blocks)
        const int64_t idx = (ro.kind & 16) ? xcd_pair(widx) : widx;
        const int64_t col = (idx % ncolblocks) * C::FPW + grp, outer = idx / ncolblocks;
        const int64_t base = outer * os + col;
        cf v[C::R];
#pragma unroll
        for (int t = 0; t < C::R; ++t) v[t] = in[base + (int64_t)(tid + C::T * t) * es];
        xf.fwd(v, lds, tid, L);
        // (32-bit indices: Ltot <= 2^27 here)
        const int n32 = (int)ro.n, L32 = (int)ro.Ltot;
        const int jb = 2 * ((int)col + (int)ncols * (int)outer), js = 2 * (int)ncols * (int)nouter;
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int j2 = jb + js * (tid + C::T * t);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int i = j2 + e;
                const float val = nrm * (e ? -v[t].y : v[t].x);
                if (i < n32) ro.co[i + n32 - 1] = val;
                else if (i > L32 - n32) ro.co[i - L32 + n32 - 1] = val;
            }
        }
    }
}

// adjoint of k_fft_cols for the reversed-order inverse: out[i] = sum_k conj(W_N^{mc k}) in[k] W_L^{-i k} (unscaled) as
// conj(fft(W_N^{mc k} conj(in[k]))).  OUT 0: complex, in place or not; OUT 2: the LAST pass of the half-length Hilbert -- element
// index j = base + i es is the sample-pair index, written is the analytic signal a[2j] = rx[2j] + i scale Re, a[2j+1] = rx[2j+1] + i
// scale Im (RowsOut kind 2's output, 16 adjacent columns = 256 contiguous bytes per row).
template <int L, int OUT>
__global__ __launch_bounds__(WgCfg<L>::WG) __attribute__((amdgpu_waves_per_eu(OUT == 0 ? SP_COLSX_WAVES : 2, OUT == 0 ? SP_COLSX_WAVES : 2))) void k_fft_cols_inv(
    const cf *__restrict__ in, cf *__restrict__ out, int64_t ncolblocks, int64_t nouter, int64_t es, int64_t os, int64_t twmul,
    XfTables tb, BigTw bt, float scale, RowsOut ro) {
    using X = XfPow2<L>;
    using C = typename X::C;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *smem = reinterpret_cast<cf *>(smem_raw);
    const int grp = C::FPW == 1 ? 0 : (int)threadIdx.x % C::FPW;
    const int tid = C::FPW == 1 ? (int)threadIdx.x : (int)threadIdx.x / C::FPW;
    cf *lds = smem + grp * C::LDS_PER;
    X xf;
    xf.init(tb, tid);
    const int64_t total = nouter * ncolblocks;
    for (int64_t idx = blockIdx.x; idx < total; idx += gridDim.x) {
        const int64_t col = (idx % ncolblocks) * C::FPW + grp;
        const int64_t base = (idx / ncolblocks) * os + col;
        cf v[C::R];
        {
            const cf *pin = in + base;
            const int ies = (int)es;
#pragma unroll
            for (int t = 0; t < C::R; ++t) {
                const cf a = pin[(tid + C::T * t) * ies];
                v[t] = mk(a.x, -a.y);
            }
        }
        const int64_t mc = twmul * col;
        if constexpr (C::R == 16) {
            bigtw_apply16(v, bt, mc * (int64_t)tid, mc * (int64_t)C::T);
        } else {
#pragma unroll
            for (int t = 0; t < C::R; ++t) {
                const int64_t m = mc * (int64_t)(tid + C::T * t);
                v[t] = cmul(v[t], cmul(bt.hi[m >> bt.lb], bt.lo[m & ((1 << bt.lb) - 1)]));
            }
        }
        xf.fwd(v, lds, tid, L);
        if constexpr (OUT == 3) {
            // OUT 2 for a full, 8-byte aligned row (ro.n = Ltot): the sample pair as one load, no predicates
            float4 *ao = reinterpret_cast<float4 *>(ro.co);
            const cf *xp = reinterpret_cast<const cf *>(ro.rx);
#pragma unroll
            for (int t = 0; t < C::R; ++t) {
                const int64_t j = base + (int64_t)(tid + C::T * t) * es;
                const cf xx = xp[j];
                ao[j] = make_float4(xx.x, scale * v[t].x, xx.y, -scale * v[t].y);
            }
        } else if constexpr (OUT == 2) {
            float4 *ao = reinterpret_cast<float4 *>(ro.co);
#pragma unroll
            for (int t = 0; t < C::R; ++t) {
                const int64_t j = base + (int64_t)(tid + C::T * t) * es;
                const bool ok0 = 2 * j < ro.n, ok1 = 2 * j + 1 < ro.n;
                const float x0 = ro.rx[ok0 ? 2 * j : 0], x1 = ro.rx[ok1 ? 2 * j + 1 : 0];
                ao[j] = make_float4(ok0 ? x0 : 0.f, scale * v[t].x, ok1 ? x1 : 0.f, -scale * v[t].y);
            }
        } else {
            cf *pout = out + base;
            const int ies = (int)es;
#pragma unroll
            for (int t = 0; t < C::R; ++t) pout[(tid + C::T * t) * ies] = mk(scale * v[t].x, -scale * v[t].y);
        }
    }
}

// ------------------------------------------------------------------------------------------
// A3+A4  fused Welch PSD (fft_analysis.py:2156-2176 loop + :1946 |X|^2 + :1980 mean), generic form.
// Each group owns frames [gid*fpg, (gid+1)*fpg); |X|^2 is accumulated in registers over the
// run, one partial spectrum per group goes to HBM.  trend[4] (device) is removed before the window
// (global detrend :2148).
// ------------------------------------------------------------------------------------------
template <class X, bool CPLX, bool LIN>
__global__ __launch_bounds__(X::C::WG) void k_welch(const void *__restrict__ x, const float *__restrict__ win, int hop,
                                                     int64_t nframes, int64_t fpg, const float *__restrict__ trend,
                                                     XfTables tb, float *__restrict__ partial, int segmean) {
    SP_KERNEL_PROLOGUE(X)
    float w[C::R], acc[C::R];
#pragma unroll
    for (int t = 0; t < C::R; ++t) {
        const int i = tid + C::T * t;
        w[t] = (X::EXACT || i < n) ? win[i] : 0.f;
        acc[t] = 0.f;
    }
    const Trend tr = load_trend(trend);
    const int64_t gid = (int64_t)blockIdx.x * C::FPW + grp;
    const int64_t g0 = gid * fpg;
    for (int64_t i = 0; i < fpg; ++i) {
        const int64_t g = g0 + i;
        // frames past the end are clamped to the last one and weighted 0: every load is unconditional
        // (a per-load predicate makes hipcc branch around each load and serialise the round trips)
        const float keep = g < nframes ? 1.f : 0.f;
        const int64_t base = (g < nframes ? g : nframes - 1) * hop;
        cf v[C::R];
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int j = tid + C::T * t;
            v[t] = load_sample(x, base + (X::EXACT || j < n ? j : n - 1), CPLX);
        }
        if (segmean) segment_detrend<C>(v, lds, tid, n, X::EXACT, segmean);   // per-segment (mlab) detrend; uniform
#pragma unroll
        for (int t = 0; t < C::R; ++t) v[t] = w[t] * detrended<LIN>(v[t], tr, base + tid + C::T * t);
        fwd_row(xf, v, lds, tid, n);
#pragma unroll
        for (int t = 0; t < C::R; ++t) acc[t] += keep * cnorm(v[t]);
    }
#pragma unroll
    for (int t = 0; t < C::R; ++t) partial[gid * X::L + tid + C::T * t] = acc[t];
}

// ------------------------------------------------------------------------------------------
// Real input, power only: two consecutive real frames ride in one complex transform, z = f_g + i f_{g+1}.
// |X_g|^2 + |X_{g+1}|^2 = (|Z[k]|^2 + |Z[n-k]|^2) / 2, so the kernel just accumulates |Z|^2 and the finish kernel
// symmetrises (k_welch_finish with `sym`).  A lone last frame has a zero imaginary part, for which the same
// formula holds.  Halves the transform count of real-valued Welch PSDs.
// ------------------------------------------------------------------------------------------
template <class X, bool LIN>
__device__ __forceinline__ void welch_rp_body(const float *__restrict__ x, const float *__restrict__ win, int hop,
                                              int64_t nframes, int64_t ppg /*pairs per group*/,
                                              const float *__restrict__ trend, XfTables tb, float *__restrict__ partial) {
    SP_KERNEL_PROLOGUE(X)
    float w[C::R], acc[C::R];
#pragma unroll
    for (int t = 0; t < C::R; ++t) {
        const int i = tid + C::T * t;
        w[t] = (X::EXACT || i < n) ? win[i] : 0.f;
        acc[t] = 0.f;
    }
    const Trend tr = load_trend(trend);
    const int64_t npairs = (nframes + 1) / 2;
    const int64_t gid = (int64_t)blockIdx.x * C::FPW + grp;
    const int64_t p0 = gid * ppg;
    for (int64_t i = 0; i < ppg; ++i) {
        const int64_t p = p0 + i;
        const float keep = p < npairs ? 1.f : 0.f;
        const int64_t ga = 2 * (p < npairs ? p : npairs - 1);
        const bool has_b = ga + 1 < nframes;
        const int64_t base_a = ga * hop, base_b = (has_b ? ga + 1 : ga) * hop;
        const float kb = has_b ? 1.f : 0.f;
        cf v[C::R];
        // (no software prefetch here: it costs 38 VGPRs = one wave per SIMD and measured no faster)
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int j = tid + C::T * t;
            const int jj = (X::EXACT || j < n) ? j : n - 1;
            v[t] = mk(x[base_a + jj], x[base_b + jj]);
        }
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int j = tid + C::T * t;
            const cf a = detrended<LIN>(mk(v[t].x, 0.f), tr, base_a + j);
            const cf b = detrended<LIN>(mk(v[t].y, 0.f), tr, base_b + j);
            v[t] = mk(w[t] * a.x, kb * w[t] * b.x);
        }
        fwd_row(xf, v, lds, tid, n);
#pragma unroll
        for (int t = 0; t < C::R; ++t) acc[t] += keep * cnorm(v[t]);
    }
#pragma unroll
    for (int t = 0; t < C::R; ++t) partial[gid * X::L + tid + C::T * t] = acc[t];
}
// entry points: plain, and with the 2-waves-per-SIMD scheduling hint (see k_welch_carry).  Without the hint the 2048-point
// kernel takes 165 VGPRs (3 waves per SIMD), with it 195 (2 waves) and is 13 % faster; 4096: -5 %, 8192: -2 %, 512: equal,
// 1024: +2 % (profiles/r01_ab_waves_per_eu.txt) -- used from 2048 points up, and not for the linear-detrend variants, which
// the hint makes spill.  The same hint on k_fft_c2c, k_stft_rp, k_fftfilt, k_pairspec and k_welch_csd_pair measured within
// noise and is not applied.
template <class X, bool LIN>
__global__ __launch_bounds__(X::C::WG) void k_welch_rp(const float *__restrict__ x, const float *__restrict__ win, int hop,
                                                        int64_t nframes, int64_t ppg, const float *__restrict__ trend,
                                                        XfTables tb, float *__restrict__ partial) {
    welch_rp_body<X, LIN>(x, win, hop, nframes, ppg, trend, tb, partial);
}
template <class X, bool LIN>
__global__ __launch_bounds__(X::C::WG) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_welch_rp_h(
    const float *__restrict__ x, const float *__restrict__ win, int hop, int64_t nframes, int64_t ppg,
    const float *__restrict__ trend, XfTables tb, float *__restrict__ partial) {
    welch_rp_body<X, LIN>(x, win, hop, nframes, ppg, trend, tb, partial);
}

// ------------------------------------------------------------------------------------------
// The metric kernel: power-of-two n = N, hop = SHIFT*T (the hop is a whole number of register
// slots), so a frame advances by renaming registers: the overlapped part of the next frame is
// carried in registers and only the SHIFT new slots per thread are read from HBM -- every sample
// is fetched once -- and those loads are issued one frame ahead of their use (software prefetch).
// The detrend constant is subtracted once per sample when it arrives.
//
// ONEPASS (global-mean detrend without a separate pass over the signal): `trend` holds an ESTIMATE
// mu0 of the mean; the kernel additionally accumulates, per group, the sum over its frames of the
// frame's last hop-block (time domain, SHIFT slots).  From those block sums the epilogue rebuilds
// sum_g X_g[k] and the exact mean and applies  |X - dW|^2 = |X|^2 - 2Re(conj(dW) X) + |dW|^2.
// ------------------------------------------------------------------------------------------
// exchange images per transform in the carry kernel: 2 = ping-pong (2 barriers per frame instead of 4, 66 KiB LDS);
// measured no faster than 1 (0.66 vs 0.64 ms at the metric shape), so the smaller footprint is kept
#ifndef SP_CARRY_NBUF
#define SP_CARRY_NBUF 1
#endif
// SP_CARRY_W3=1 (experiment): the carry kernel at 3 waves per SIMD (<= 168 VGPRs): the pass-1 twiddle constants live in a
// 2.8 KiB LDS table (40 floats re-read per thread and frame) and the one-pass block sums are accumulated by ds_add_f32 in
// LDS (16 KiB per workgroup at hop 2048) instead of 16 VGPRs + 16 VALU adds
#ifndef SP_CARRY_W3
#define SP_CARRY_W3 0
#endif
#ifndef SP_W3_SUMS           /* 1: block sums by LDS atomics (with SP_CARRY_W3) */
#define SP_W3_SUMS 1
#endif
#ifndef SP_W3_BOUND          /* waves per SIMD the W3 kernel is compiled for */
#define SP_W3_BOUND 3
#endif
#ifndef SP_W3_TW             /* 1: pass-1 constants from the LDS table (with SP_CARRY_W3) */
#define SP_W3_TW 1
#endif
// SP_EST_IN_KERNEL=1 (default): the one-pass kernel estimates mu0 itself (no k_op_estimate launch)
#ifndef SP_EST_IN_KERNEL
#define SP_EST_IN_KERNEL 1
#endif
// SP_WELCH_PIPE_DEFAULT: 1 = nfft 4096 one-pass Welch goes through k_welch_pipe (k_welch_pipe.hip); SP_WELCH_PIPE=0/1 overrides at run time
#ifndef SP_WELCH_PIPE_DEFAULT
#define SP_WELCH_PIPE_DEFAULT 1
#endif
// SP_NT_LOADS=1: the carry kernel's sample loads carry the non-temporal cache policy
#ifndef SP_NT_LOADS
#define SP_NT_LOADS 0
#endif
// (no min-waves hint: capping at 168 VGPRs makes hipcc spill the window registers and reload them inside the
//  frame loop behind vmcnt(0) waits, which also drains the prefetch loads -- measured 2x slower)
// COG: instead of accumulating |X|^2 over the frames, every frame's spectral moments sum ks|X|^2, sum |X|^2 (signed bin
// index ks, every bin) are reduced across the wave and stored in the slot (frame, wave) of a float2 array
// (`partial` carries that pointer; k_cog_finish adds the slots) -- the centre of gravity per frame of Doppler.cog / cogspec (Doppler.py:43-81)
// on the streaming path of the metric kernel.
template <int N, bool CPLX, int SHIFT, bool ONEPASS, bool COG>
__device__ __forceinline__ void welch_carry_body(
    const void *__restrict__ x, const float *__restrict__ win, int64_t nframes, int64_t fpg,
    const float *__restrict__ trend, XfTables tb, float *__restrict__ partial, cf *__restrict__ spartial) {
    constexpr bool W3 = SP_CARRY_W3 && ONEPASS && !COG && N == 4096 && WgCfg<N>::FPW == 1;
    constexpr bool W3S = W3 && SP_W3_SUMS;
    using X = XfPow2<N, W3 && SP_W3_TW>;
    SP_KERNEL_PROLOGUE(X)
    (void)n;
    // W3: after the exchange image(s): block sums [SHIFT][2][T] floats, then the pass-1 constant table
    float *lds_sums = reinterpret_cast<float *>(smem + C::FPW * C::LDS_PER * SP_CARRY_NBUF);
    if constexpr (W3) {
        float *tab = lds_sums + 2 * SHIFT * C::T;
#pragma unroll
        for (int s = 0; s < 2 * SHIFT; ++s) lds_sums[s * C::T + tid] = 0.f;
        xf.f.publish_tw1(tab, tid);
        __syncthreads();
    }
    static_assert(!(COG && ONEPASS), "the moments mode has no one-pass detrend epilogue");
    static_assert(SHIFT >= 1 && SHIFT <= C::R, "hop must be 1..R register slots");
    constexpr int KEEP = C::R - SHIFT;
    constexpr bool UNI = C::FPW == 1;          // one group per workgroup: trip counts may differ between groups
    const int hop = SHIFT * C::T;
    float w[C::R], acc[C::R];
    cf sacc[SHIFT];
#pragma unroll
    for (int t = 0; t < C::R; ++t) {
        w[t] = win[tid + C::T * t];
        acc[t] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < SHIFT; ++s) sacc[s] = mk(0.f, 0.f);
    cf mu;
    if constexpr (ONEPASS && SP_EST_IN_KERNEL) {
        // the mean ESTIMATE mu0 of the one-pass detrend, computed by every workgroup for itself from the same 16 runs of WG
        // samples spread over the frames' span (identical loads and reduction order in every workgroup -> identical mu0;
        // they hit L2 after the first workgroup): saves the separate k_op_estimate launch (12 us + a launch gap per
        // step).  Any mu0 gives the exact result (the epilogue corrects with the true mean); workgroup 0 publishes it in
        // trend[] for the epilogue kernels.
        const int64_t span = (nframes - 1) * (int64_t)hop + N;
        const int64_t pitch = span / 16;
        float sx = 0.f, sy = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int64_t i = pitch * r + (int64_t)threadIdx.x;
            i = i < span ? i : span - 1;
            const cf v = load_sample(x, i, CPLX);
            sx += v.x;
            sy += v.y;
        }
        sx = wave_sum64(sx);
        sy = wave_sum64(sy);
        float *red = reinterpret_cast<float *>(smem);            // before the first transform uses the exchange image
        if ((threadIdx.x & 63) == 0) {
            red[2 * (threadIdx.x >> 6)] = sx;
            red[2 * (threadIdx.x >> 6) + 1] = sy;
        }
        __syncthreads();
        double tx = 0.0, ty = 0.0;
#pragma unroll
        for (int wv = 0; wv < C::WG / 64; ++wv) {
            tx += (double)red[2 * wv];
            ty += (double)red[2 * wv + 1];
        }
        const double cnt = 16.0 * (double)C::WG;
        mu = mk((float)(tx / cnt), (float)(ty / cnt));
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            float *tw = const_cast<float *>(trend);
            tw[0] = mu.x;
            tw[1] = mu.y;
            tw[2] = 0.f;
            tw[3] = 0.f;
        }
    } else {
        mu = load_trend(trend).m;
    }
    // keep the constant in VGPRs: a VALU instruction with an SGPR source issues at half rate on gfx950
    // (tools/ubench/vgpr_bank.hip), and it is subtracted from every arriving sample
    asm volatile("" : "+v"(mu.x), "+v"(mu.y));
    const int64_t gid = (int64_t)blockIdx.x * C::FPW + grp;
    const int64_t g0 = gid * fpg;
    const int64_t last = nframes - 1;
    int64_t trips = fpg;
    if (UNI) {
        trips = nframes - g0;
        trips = trips < 0 ? 0 : (trips > fpg ? fpg : trips);
    }
    cf raw[C::R];
    {
        const int64_t base = (g0 < nframes ? g0 : last) * hop + tid;
#pragma unroll
        for (int t = 0; t < C::R; ++t) raw[t] = load_sample(x, base + C::T * t, CPLX);
#pragma unroll
        for (int t = 0; t < C::R; ++t) raw[t] = raw[t] - mu;
    }
    // new slots of frame gq (clamped at the end of the signal; unused then)
    auto issue = [&](cf (&dst)[SHIFT], int64_t gq) __attribute__((always_inline)) {
        int64_t gn = gq < nframes ? gq : last;
        if constexpr (SP_ABLATE & 16) gn = blockIdx.x & 7;          // diagnostic: every load hits L2 (8 distinct frames)
        if constexpr (UNI && !(SP_ABLATE & 8)) {
            // one group per workgroup: the frame base is uniform -> scalar base pointer + 32-bit lane offset
            const int64_t ubase = gn * hop + (int64_t)C::T * KEEP;
#pragma unroll
            for (int s = 0; s < SHIFT; ++s) {
                const unsigned off = (unsigned)(tid + C::T * s);
#if SP_NT_LOADS
                // the stream is read exactly once: non-temporal loads (nt) keep it from displacing the twiddle / window
                // tables and the partial spectra in L2
                if (CPLX) {
                    const sp_f2v r = __builtin_nontemporal_load(reinterpret_cast<const sp_f2v *>(x) + ubase + off);
                    dst[s] = mk(r.x, r.y);
                } else {
                    dst[s] = mk(__builtin_nontemporal_load(reinterpret_cast<const float *>(x) + ubase + off), 0.f);
                }
#else
                if (CPLX) dst[s] = (reinterpret_cast<const cf *>(x) + ubase)[off];
                else dst[s] = mk((reinterpret_cast<const float *>(x) + ubase)[off], 0.f);
#endif
            }
        } else {
            const int64_t base = gn * hop + tid + (int64_t)C::T * KEEP;
#pragma unroll
            for (int s = 0; s < SHIFT; ++s) {
                if constexpr (SP_ABLATE & 8) dst[s] = raw[s] + mu;  // diagnostic: no global loads in the loop
                else dst[s] = load_sample(x, base + C::T * s, CPLX);
            }
        }
    };
    // COG: a wave keeps the moments of its last W frames spread over its lanes (frame i in lane i mod W) and writes them
    // with one coalesced store every W frames, layout acc[wave of the frame][frame], summed over the waves by k_cog_finish.
    // (First form: two float64 atomicAdd per wave and frame after __shfl_xor sums -- 0.86 ms kernel at the metric shape;
    //  per-wave slots with plain stores, DPP sums and the literal-FMA moments below: 0.71-0.79 ms.)
    cf pend = mk(0.f, 0.f);
    auto flush = [&](int64_t i) __attribute__((always_inline)) {
        constexpr int W = C::T < 64 ? C::T : 64;
        const int sel = (int)(i & (W - 1)), li = tid & (W - 1);
        const int64_t gg = g0 + (i - sel) + li;
        if (li <= sel && gg < nframes) reinterpret_cast<cf *>(partial)[(int64_t)(tid / W) * nframes + gg] = pend;
    };
    // one frame: window, transform, accumulate; `fill` receives a prefetch (issued once v is formed, so that the
    // incoming samples can take over the registers of the slots that just died), `take` holds the new slots of frame g+1
    auto body = [&](int64_t i, cf (&fill)[SHIFT], int64_t fill_frame, cf (&take)[SHIFT]) __attribute__((always_inline)) {
        const int64_t g = g0 + i;
        const float keep = (UNI || g < nframes) ? 1.f : 0.f;
        if constexpr (W3S) {
            // own slots, plain read-modify-write (ds_add_f32 measured ~300 cycles per wave-instruction: 3.1 ms kernel)
            cf *ss = reinterpret_cast<cf *>(lds_sums);
            cf cur[SHIFT];
#pragma unroll
            for (int s = 0; s < SHIFT; ++s) cur[s] = ss[s * C::T + tid];
#pragma unroll
            for (int s = 0; s < SHIFT; ++s) ss[s * C::T + tid] = cur[s] + raw[KEEP + s];
        } else if (ONEPASS) {
#pragma unroll
            for (int s = 0; s < SHIFT; ++s) sacc[s] = UNI ? sacc[s] + raw[KEEP + s] : sacc[s] + keep * raw[KEEP + s];
        }
        cf v[C::R];
#pragma unroll
        for (int t = 0; t < C::R; ++t) v[t] = w[t] * raw[t];
        issue(fill, fill_frame);
        if (SP_CARRY_NBUF == 2) {
            // ping-pong exchange images: with an odd number of exchanges per transform the roles swap every frame
            cf *lds_b = lds + C::FPW * C::LDS_PER;
            const bool swap = ((C::PL::NP - 1) & 1) && (i & 1);
            xf.fwd2(v, swap ? lds_b : lds, swap ? lds : lds_b, tid);
        } else {
            if constexpr (SP_CARRY_WAVELOCAL && C::T <= 64 && C::FPW > 1) xf.fwdw(v, lds, tid, N);
            else xf.fwd(v, lds, tid, N);
        }
        if constexpr (COG) {
            // every bin (Doppler.cog's form; a band limit goes through the generic kernel): with ks = tid + c_t,
            // c_t = T t - (N in the upper half), sum ks p = tid * sum p + sum c_t p -- one FMA with a literal per bin
            float numc = 0.f, den = 0.f;
#pragma unroll
            for (int t = 0; t < C::R; ++t) {
                const float p = cnorm(v[t]);
                numc = fmaf(p, (float)(C::T * t - (t >= C::R / 2 ? N : 0)), numc);
                den += p;
            }
            float num = fmaf((float)tid, den, numc);
            constexpr int W = C::T < 64 ? C::T : 64;
            if constexpr (W == 64) {
                num = wave_sum64(num);
                den = wave_sum64(den);
            } else {
#pragma unroll
                for (int o = W / 2; o > 0; o >>= 1) {
                    num += __shfl_xor(num, o);
                    den += __shfl_xor(den, o);
                }
            }
            if ((tid & (W - 1)) == (int)(i & (W - 1))) pend = mk(num, den);
            if ((i & (W - 1)) == W - 1 || i == trips - 1) flush(i);
        } else {
#pragma unroll
            for (int t = 0; t < C::R; ++t) {
                // two chained FMAs per bin (`acc + (x*x + y*y)` would be mul + fma + add)
                if (UNI) acc[t] = fmaf(v[t].y, v[t].y, fmaf(v[t].x, v[t].x, acc[t]));
                else acc[t] += keep * cnorm(v[t]);
            }
        }
        // advance one hop: rename registers, detrend the samples that arrived
#pragma unroll
        for (int t = 0; t < KEEP; ++t) raw[t] = raw[t + SHIFT];
#pragma unroll
        for (int s = 0; s < SHIFT; ++s) raw[KEEP + s] = take[s] - mu;
    };
    // (prefetching two frames ahead -- a ring of two buffers, the loop unrolled by two -- measured no faster: the cost
    //  of streaming from HBM, 0.52 -> 0.64 ms against the same loads hitting L2, is not exposed latency)
    for (int64_t i = 0; i < trips; ++i) {
        cf nx[SHIFT];
        body(i, nx, g0 + i + 1, nx);
    }
    if constexpr (!COG) {
#pragma unroll
        for (int t = 0; t < C::R; ++t) partial[gid * N + tid + C::T * t] = acc[t];
    }
    if constexpr (W3S) {
#pragma unroll
        for (int s = 0; s < SHIFT; ++s)
            spartial[gid * hop + tid + C::T * s] = reinterpret_cast<cf *>(lds_sums)[s * C::T + tid];
    } else if (ONEPASS) {
#pragma unroll
        for (int s = 0; s < SHIFT; ++s) spartial[gid * hop + tid + C::T * s] = sacc[s];
    }
}

// The two entry points of the body above.  The Welch kernel tells the compiler that it runs at exactly 2 waves per SIMD
// (amdgpu_waves_per_eu): at 215 VGPRs the next occupancy step (168) is out of reach, and without the hint the scheduler
// still trades latency hiding for register pressure -- with it: 205 VGPRs and 3 % less time at the metric shape
// (0.708 vs 0.731 ms on the same box, three interleaved rounds).  The moments kernel measured 2 % slower with the hint
// and stays without.
#if SP_CARRY_W3
template <int N, bool CPLX, int SHIFT, bool ONEPASS>
__global__ __launch_bounds__(WgCfg<N>::WG, (ONEPASS && N == 4096) ? SP_W3_BOUND : 2) void k_welch_carry(
    const void *__restrict__ x, const float *__restrict__ win, int64_t nframes, int64_t fpg,
    const float *__restrict__ trend, XfTables tb, float *__restrict__ partial, cf *__restrict__ spartial) {
    welch_carry_body<N, CPLX, SHIFT, ONEPASS, false>(x, win, nframes, fpg, trend, tb, partial, spartial);
}
#else
template <int N, bool CPLX, int SHIFT, bool ONEPASS>
__global__ __launch_bounds__(WgCfg<N>::WG) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_welch_carry(
    const void *__restrict__ x, const float *__restrict__ win, int64_t nframes, int64_t fpg,
    const float *__restrict__ trend, XfTables tb, float *__restrict__ partial, cf *__restrict__ spartial) {
    welch_carry_body<N, CPLX, SHIFT, ONEPASS, false>(x, win, nframes, fpg, trend, tb, partial, spartial);
}
#endif
// (without the hint: the 256-point variants fit 3 waves per SIMD, 148-168 VGPRs, and must not be capped at 2)
template <int N, bool CPLX, int SHIFT, bool ONEPASS>
__global__ __launch_bounds__(WgCfg<N>::WG) void k_welch_carry_nh(
    const void *__restrict__ x, const float *__restrict__ win, int64_t nframes, int64_t fpg,
    const float *__restrict__ trend, XfTables tb, float *__restrict__ partial, cf *__restrict__ spartial) {
    welch_carry_body<N, CPLX, SHIFT, ONEPASS, false>(x, win, nframes, fpg, trend, tb, partial, spartial);
}
template <int N, bool CPLX, int SHIFT>
__global__ __launch_bounds__(WgCfg<N>::WG) void k_welch_carry_cog(
    const void *__restrict__ x, const float *__restrict__ win, int64_t nframes, int64_t fpg,
    const float *__restrict__ trend, XfTables tb, cf *__restrict__ slots) {
    welch_carry_body<N, CPLX, SHIFT, false, true>(x, win, nframes, fpg, trend, tb, reinterpret_cast<float *>(slots), nullptr);
}

// ---- one-pass detrend epilogue (all tiny, double precision) -------------------------------------
// state layout (doubles): A[N] | Sl[2*H] | tot[2] = sum_{i<nmean}(x[i]-mu0) | dlt[2] = mean - mu0 | cnt[1]
struct OnePass {
    double *A, *Sl, *tot, *dlt;
    int sym = 0;       // 1: A holds sum |Z|^2 of real-pair transforms (Z = X_2q + i X_2q+1): sum |X|^2 [k] = (A[k] + A[N-k]) / 2
};

// column sums, in double, of two float matrices with G rows in ONE launch: m0[G][c0] -> o0[c0] (the raw |X|^2 sums
// A[k]) and m1[G][c1] -> o1[c1] (the block sums: spartial[G][H] complex seen as [G][2H] floats -> Sl[2j], Sl[2j+1]).
// block = 32 columns x 32 row slices (1024 threads), 4 independent loads in flight per thread; deterministic order.
static __global__ __launch_bounds__(1024) void k_op_colsums(const float *__restrict__ m0, int c0, double *__restrict__ o0,
                                                             const float *__restrict__ m1, int c1, double *__restrict__ o1,
                                                             int64_t G) {
    __shared__ double sh[32][32];
    const int nb0 = (c0 + 31) / 32;
    const bool second = (int)blockIdx.x >= nb0;
    const float *__restrict__ m = second ? m1 : m0;
    const int cols = second ? c1 : c0;
    double *__restrict__ o = second ? o1 : o0;
    const int lane = threadIdx.x % 32, sl = threadIdx.x / 32;
    const int k = ((int)blockIdx.x - (second ? nb0 : 0)) * 32 + lane;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (k < cols) {
        int64_t g = sl;
        for (; g + 96 < G; g += 128) {
            const float a0 = m[g * cols + k], a1 = m[(g + 32) * cols + k], a2 = m[(g + 64) * cols + k], a3 = m[(g + 96) * cols + k];
            s0 += (double)a0;
            s1 += (double)a1;
            s2 += (double)a2;
            s3 += (double)a3;
        }
        for (; g < G; g += 32) s0 += (double)m[g * cols + k];
    }
    sh[sl][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && k < cols) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 32; ++q) t += sh[q][lane];
        o[k] = t;
    }
}

// tot = sum_{i < nmean} (x[i] - mu0): block sums cover [(r-1)H, (M+r-1)H); add the head blocks and fix the end.
// one block of 1024 threads.
// also: sum_out = tot + nmean*mu0 (the shard's plain sample sum) and dlt = tot/nmean (delta for the shard's own mean)
template <bool CPLX>
static __global__ __launch_bounds__(1024) void k_op_total(const void *__restrict__ x, const float *__restrict__ trend,
                                                           const double *__restrict__ Sl, int H, int r, int64_t M,
                                                           int64_t nmean, double *__restrict__ tot,
                                                           double *__restrict__ dlt, double *__restrict__ sum_out) {
    __shared__ double sh[2][1024];
    const cf mu = mk(trend[0], trend[1]);
    double a = 0, b = 0;
    for (int j = threadIdx.x; j < H; j += 1024) {
        a += Sl[2 * j];
        b += Sl[2 * j + 1];
    }
    const int64_t head = (int64_t)(r - 1) * H;           // samples before the first counted block
    const int64_t cov = (M + r - 1) * (int64_t)H;        // end of the last counted block
    for (int64_t i = threadIdx.x; i < head; i += 1024) {
        const cf v = load_sample(x, i, CPLX) - mu;
        a += v.x;
        b += v.y;
    }
    if (nmean > cov) {
        for (int64_t i = cov + threadIdx.x; i < nmean; i += 1024) {
            const cf v = load_sample(x, i, CPLX) - mu;
            a += v.x;
            b += v.y;
        }
    } else {
        for (int64_t i = nmean + threadIdx.x; i < cov; i += 1024) {
            const cf v = load_sample(x, i, CPLX) - mu;
            a -= v.x;
            b -= v.y;
        }
    }
    sh[0][threadIdx.x] = a;
    sh[1][threadIdx.x] = b;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        tot[0] = sh[0][0];
        tot[1] = sh[1][0];
        dlt[0] = sh[0][0] / (double)nmean;
        dlt[1] = sh[1][0] / (double)nmean;
        sum_out[0] = sh[0][0] + (double)nmean * (double)trend[0];
        sum_out[1] = sh[1][0] + (double)nmean * (double)trend[1];
    }
}

// one workgroup: c[n] = sum_g x_g[n] rebuilt from the block sums and the few edge blocks, B = FFT(w c) = sum_g X_g,
// then out[slot] = scale * doubling * (A[k] - 2 Re(conj(d W[k]) B[k]) + M |d W[k]|^2) with d = mean - mu0.
// mean_in != null (the caller's global mean) overrides the shard's own delta.
// EXPORT: instead of the finished spectrum, write this shard's additive state (see sp_welch_export) into `out`:
//   out[0..N) = A[k]   out[N..3N) = B[k] (re, im)   out[3N..5N) = conj(mu0) B[k]
//   out[5N..5N+8) = M mu0 (re, im), M |mu0|^2, sum of the nmean own samples (re, im), M, nmean, 0
template <int N, bool CPLX, bool EXPORT = false>
static __global__ __launch_bounds__(WgCfg<N>::WG) void k_op_finish(const void *__restrict__ x, const float *__restrict__ trend,
                                                                    const float *__restrict__ win,
                                                                    const double *__restrict__ Sl,
                                                                    const double *__restrict__ A, const cf *__restrict__ Wf,
                                                                    const double *__restrict__ dlt_local,
                                                                    const double *__restrict__ mean_in, int H, int r,
                                                                    int64_t M, int64_t nmean, int sided, double scale,
                                                                    XfTables tb, double *__restrict__ out, int sym,
                                                                    int64_t x_cs = 0, int64_t sl_cs = 0, int64_t out_cs = 0) {
    using X = XfPow2<N>;
    SP_KERNEL_PROLOGUE(X)
    (void)n;
    // one workgroup per signal: blockIdx.x selects the channel of a multi-channel call (strides in samples / doubles; all zero
    // for the single-signal callers, whose grid is one workgroup)
    x = reinterpret_cast<const char *>(x) + (int64_t)blockIdx.x * x_cs * (CPLX ? 8 : 4);
    trend += x_cs ? 4 * blockIdx.x : 0;
    Sl += (int64_t)blockIdx.x * sl_cs;
    out += (int64_t)blockIdx.x * out_cs;
    const cf mu = mk(trend[0], trend[1]);
    // this single workgroup is one chain of memory round trips: everything that does not depend on a computed value
    // (window, raw sums, FFT(window)) is fetched up front, together with the twiddle tables of the prologue
    double a_pre[C::R];
    cf wf_pre[C::R];
    float win_pre[C::R];
#pragma unroll
    for (int t = 0; t < C::R; ++t) {
        const int k = tid + C::T * t;
        a_pre[t] = sym ? 0.5 * (A[k] + A[(N - k) & (N - 1)]) : A[k];
        wf_pre[t] = Wf[k];
        win_pre[t] = win[k];
    }
    double dr, di;
    double tot_r = 0.0, tot_i = 0.0;
    if (!EXPORT && mean_in) {
        dr = mean_in[0] - (double)trend[0];
        di = mean_in[1] - (double)trend[1];
    } else if (!EXPORT && dlt_local) {
        dr = dlt_local[0];
        di = dlt_local[1];
    } else {
        // the shard's own mean (what k_op_total computes for the split ABI), here without the extra launch:
        // sum_{i<nmean}(x[i] - mu0) = all block sums + head blocks +/- the ragged end
        double a = 0, b = 0;
        // H <= N and head < N: fixed trip counts with masks, so that all loads of a thread are in flight together
        // (this workgroup is one latency chain; with data-dependent loops it cost 28 us per step)
        constexpr int NIT = N / C::WG > 0 ? N / C::WG : 1;
        const int64_t head = (int64_t)(r - 1) * H, cov = (M + r - 1) * (int64_t)H;
        {
            double sa[NIT], sb[NIT];
            cf hv[NIT];
#pragma unroll
            for (int q = 0; q < NIT; ++q) {
                const int j = (int)threadIdx.x + q * C::WG;
                const int jc = j < H ? j : 0;
                sa[q] = Sl[2 * jc];
                sb[q] = Sl[2 * jc + 1];
                hv[q] = load_sample(x, j < head ? j : 0, CPLX);
            }
#pragma unroll
            for (int q = 0; q < NIT; ++q) {
                const int j = (int)threadIdx.x + q * C::WG;
                if (j < H) {
                    a += sa[q];
                    b += sb[q];
                }
                if (j < head) {
                    a += (double)(hv[q].x - mu.x);
                    b += (double)(hv[q].y - mu.y);
                }
            }
        }
        const int64_t lo = nmean > cov ? cov : nmean, hi = nmean > cov ? nmean : cov;
        const double sgn = nmean > cov ? 1.0 : -1.0;
        for (int64_t i = lo + threadIdx.x; i < hi; i += C::WG) {
            const cf s = load_sample(x, i, CPLX) - mu;
            a += sgn * s.x;
            b += sgn * s.y;
        }
        double *red = reinterpret_cast<double *>(smem);           // 2 x WG doubles, before the transform uses the LDS
        red[threadIdx.x] = a;
        red[C::WG + threadIdx.x] = b;
        __syncthreads();
        for (int o = C::WG / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) {
                red[threadIdx.x] += red[threadIdx.x + o];
                red[C::WG + threadIdx.x] += red[C::WG + threadIdx.x + o];
            }
            __syncthreads();
        }
        tot_r = red[0];
        tot_i = red[C::WG];
        dr = tot_r / (double)nmean;
        di = tot_i / (double)nmean;
        __syncthreads();
    }
    cf v[C::R];
#pragma unroll
    for (int t = 0; t < C::R; ++t) {
        const int nidx = tid + C::T * t;
        const int q = nidx / H, j = nidx % H;
        double a = Sl[2 * j], b = Sl[2 * j + 1];
        for (int bb = q; bb <= r - 2; ++bb) {
            const cf s = load_sample(x, (int64_t)bb * H + j, CPLX) - mu;
            a += s.x;
            b += s.y;
        }
        for (int64_t bb = M + q; bb <= M + r - 2; ++bb) {
            const cf s = load_sample(x, bb * H + j, CPLX) - mu;
            a -= s.x;
            b -= s.y;
        }
        const double wn = (double)win_pre[t];
        v[t] = (grp == 0) ? mk((float)(wn * a), (float)(wn * b)) : mk(0.f, 0.f);
    }
    xf.fwd(v, lds, tid, N);
    if constexpr (EXPORT) {
        if (grp == 0) {
            const double mr = (double)mu.x, mi = (double)mu.y;
#pragma unroll
            for (int t = 0; t < C::R; ++t) {
                const int k = tid + C::T * t;
                const double br = (double)v[t].x, bi = (double)v[t].y;
                out[k] = a_pre[t];
                out[N + 2 * k] = br;
                out[N + 2 * k + 1] = bi;
                out[3 * N + 2 * k] = mr * br + mi * bi;          // conj(mu0) B
                out[3 * N + 2 * k + 1] = mr * bi - mi * br;
            }
            if (threadIdx.x == 0) {
                double *sc = out + 5 * N;
                sc[0] = (double)M * mr;
                sc[1] = (double)M * mi;
                sc[2] = (double)M * (mr * mr + mi * mi);
                sc[3] = tot_r + (double)nmean * mr;
                sc[4] = tot_i + (double)nmean * mi;
                sc[5] = (double)M;
                sc[6] = (double)nmean;
                sc[7] = 0.0;
            }
        }
        return;
    }
    if (grp == 0) {
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int k = tid + C::T * t;
            const int slot = bin_slot(k, N, sided);
            if (slot < 0) continue;
            const double wr = wf_pre[t].x, wi = wf_pre[t].y;
            const double er = dr * wr - di * wi, ei = dr * wi + di * wr;       // d * Wf[k]
            const double p = a_pre[t] - 2.0 * (er * (double)v[t].x + ei * (double)v[t].y) + (double)M * (er * er + ei * ei);
            out[slot] = p * scale * (bin_doubled(k, N, sided) ? 2.0 : 1.0);
        }
    }
}

// The all-reduced (summed over shards) state of k_op_finish<EXPORT> -> the PSD of the whole stream, detrended by the
// global mean mu = S / n:  P[k] = A - 2 Re(conj(W) (conj(mu) B - C)) + |W|^2 (|mu|^2 M - 2 Re(conj(mu) S1) + S2)
// (each shard's sum |X - (mu - mu0_r) W|^2, expanded so that only sums over shards appear).
static __global__ __launch_bounds__(256) void k_op_apply(const double *__restrict__ st, const cf *__restrict__ Wf, int n,
                                                         int sided, double scale, double *__restrict__ out) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int slot = bin_slot(k, n, sided);
    if (slot < 0) return;
    const double *sc = st + 5 * (int64_t)n;
    const double Mt = sc[5], nt = sc[6];
    const double mr = sc[3] / nt, mi = sc[4] / nt;
    const double br = st[n + 2 * k], bi = st[n + 2 * k + 1], cr = st[3 * n + 2 * k], ci = st[3 * n + 2 * k + 1];
    const double dr = mr * br + mi * bi - cr, di = mr * bi - mi * br - ci;            // conj(mu) B - C
    const double wr = Wf[k].x, wi = Wf[k].y;
    const double cross = wr * dr + wi * di;                                            // Re(conj(W) D)
    const double s = (mr * mr + mi * mi) * Mt - 2.0 * (mr * sc[0] + mi * sc[1]) + sc[2];
    const double p = st[k] - 2.0 * cross + (wr * wr + wi * wi) * s;
    out[slot] = p * scale * (bin_doubled(k, n, sided) ? 2.0 : 1.0);
}

// mean estimate mu0: 64 contiguous runs of <= 1024 samples spread over the whole signal (robust to drift, and each
// run is a coalesced read); one block of 1024 threads writes trend[4] = (mu0, 0 slope).
#define SP_EST_RUNS 64
#define SP_EST_LEN 1024
template <bool CPLX>
static __global__ __launch_bounds__(1024) void k_op_estimate(const void *__restrict__ x, int64_t nsig,
                                                              float *__restrict__ trend) {
    __shared__ double sh[2][1024];
    const int64_t len = nsig / SP_EST_RUNS < SP_EST_LEN ? nsig / SP_EST_RUNS : SP_EST_LEN;     // may be 0 for tiny signals
    const int64_t pitch = nsig / SP_EST_RUNS;
    double a = 0, b = 0;
    // wave w covers runs w, w+16, w+32, w+48; lane l the elements l + 64 j of a run.  All 64 loads of a thread are
    // independent and unconditional (index clamped, value masked) so that they are in flight together: the kernel
    // costs one memory round trip instead of sixteen.
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (len > 0) {
#pragma unroll
        for (int m = 0; m < SP_EST_RUNS / 16; ++m) {
            const int64_t base = pitch * (wv + 16 * m);
            cf v[SP_EST_LEN / 64];
#pragma unroll
            for (int j = 0; j < SP_EST_LEN / 64; ++j) {
                const int64_t i = lane + 64 * j;
                v[j] = load_sample(x, base + (i < len ? i : len - 1), CPLX);
            }
            float fa = 0.f, fb = 0.f;          // 16 terms per partial: float is ample, the rest is summed in double
#pragma unroll
            for (int j = 0; j < SP_EST_LEN / 64; ++j) {
                const float keep = (lane + 64 * j) < len ? 1.f : 0.f;
                fa = fmaf(keep, v[j].x, fa);
                fb = fmaf(keep, v[j].y, fb);
            }
            a += (double)fa;
            b += (double)fb;
        }
    }
    sh[0][threadIdx.x] = a;
    sh[1][threadIdx.x] = b;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double cnt = (double)(len * SP_EST_RUNS);
        trend[0] = cnt > 0 ? (float)(sh[0][0] / cnt) : 0.f;
        trend[1] = cnt > 0 ? (float)(sh[1][0] / cnt) : 0.f;
        trend[2] = 0.f;
        trend[3] = 0.f;
    }
}

// sum partial[G][L] over G in double, apply sidedness + scale -> out[nbins] (double).
// block = 32 bins x 32 slices of the group range (1024 threads), 4 independent loads in flight per thread
// (the reduction is latency-bound); deterministic order.
#define SP_FIN_BINS 32
#define SP_FIN_SLICES 32
// sym != 0 (real-pair kernels): the bin sum is (S[k] + S[(n-k) % n]) / 2.
// sum over the slices of the finish kernels' [NJ][slices][bins] image by halving (fixed order: deterministic); the result of
// column `lane` is left in sh[j][0][lane].  (The first form let the 32 threads of slice 0 add 32 values each, fully unrolled:
// k_csd_pair_finish spilled 268 VGPRs and took 0.39 ms for 100 MB of partial spectra.)
template <int NJ>
__device__ __forceinline__ void fin_reduce(double (&sh)[NJ][SP_FIN_SLICES][SP_FIN_BINS], int sl, int lane) {
    for (int off = SP_FIN_SLICES / 2; off > 0; off >>= 1) {
        __syncthreads();
        if (sl < off) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) sh[j][sl][lane] += sh[j][sl + off][lane];
        }
    }
    __syncthreads();
}

static __global__ __launch_bounds__(SP_FIN_BINS *SP_FIN_SLICES) void k_welch_finish(const float *__restrict__ partial, int64_t G,
                                                                                     int L, int n, int sided, double scale,
                                                                                     double *__restrict__ out, int sym) {
    __shared__ double sh[SP_FIN_SLICES][SP_FIN_BINS];
    const int lane = threadIdx.x % SP_FIN_BINS, sl = threadIdx.x / SP_FIN_BINS;
    const int k = blockIdx.x * SP_FIN_BINS + lane;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (k < n) {
        int64_t g = sl;
        for (; g + 3 * SP_FIN_SLICES < G; g += 4 * SP_FIN_SLICES) {
            const float a0 = partial[g * L + k], a1 = partial[(g + SP_FIN_SLICES) * L + k];
            const float a2 = partial[(g + 2 * SP_FIN_SLICES) * L + k], a3 = partial[(g + 3 * SP_FIN_SLICES) * L + k];
            s0 += (double)a0;
            s1 += (double)a1;
            s2 += (double)a2;
            s3 += (double)a3;
        }
        for (; g < G; g += SP_FIN_SLICES) s0 += (double)partial[g * L + k];
        if (sym) {
            const int km = k == 0 ? 0 : n - k;
            double m = 0.0;
            for (int64_t g2 = sl; g2 < G; g2 += SP_FIN_SLICES) m += (double)partial[g2 * L + km];
            s0 = 0.5 * ((s0 + s1) + (s2 + s3) + m);
            s1 = s2 = s3 = 0.0;
        }
    }
    sh[sl][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && k < n) {
        const int slot = bin_slot(k, n, sided);
        if (slot >= 0) {
            double tot = 0.0;
#pragma unroll
            for (int j = 0; j < SP_FIN_SLICES; ++j) tot += sh[j][lane];
            out[slot] = tot * scale * (bin_doubled(k, n, sided) ? 2.0 : 1.0);
        }
    }
}

// ------------------------------------------------------------------------------------------
// A5  fft_pwelch core (fft_analysis.py:362-393): reference x against channel y_c.
// grid.y = channel.  partial layout per (channel, group): [4][L] = |X|^2, |Y|^2, Re, Im of Y conj(X)
// trend_x[4], trend_y[nch][4]
// ------------------------------------------------------------------------------------------
template <class X, bool CPLX, bool LIN>
__global__ __launch_bounds__(X::C::WG) void k_welch_csd(const void *__restrict__ x, const void *__restrict__ y,
                                                         int64_t y_ld, const float *__restrict__ win, int hop,
                                                         int64_t nframes, int64_t fpg, const float *__restrict__ trend_x,
                                                         const float *__restrict__ trend_y, XfTables tb,
                                                         float *__restrict__ partial, int64_t groups_total, int segmean) {
    SP_KERNEL_PROLOGUE(X)
    const int ch = blockIdx.y;
    float w[C::R], axx[C::R], ayy[C::R];
    cf axy[C::R];
#pragma unroll
    for (int t = 0; t < C::R; ++t) {
        const int i = tid + C::T * t;
        w[t] = (X::EXACT || i < n) ? win[i] : 0.f;
        axx[t] = ayy[t] = 0.f;
        axy[t] = mk(0.f, 0.f);
    }
    const Trend trx = load_trend(trend_x), try_ = load_trend(trend_y + 4 * ch);
    const int64_t yoff = (int64_t)ch * y_ld;
    const int64_t gid = (int64_t)blockIdx.x * C::FPW + grp;
    const int64_t g0 = gid * fpg;
    for (int64_t i = 0; i < fpg; ++i) {
        const int64_t g = g0 + i;
        const float keep = g < nframes ? 1.f : 0.f;
        const int64_t base = (g < nframes ? g : nframes - 1) * hop;
        cf vx[C::R], vy[C::R];
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int j = tid + C::T * t;
            const int64_t idx = base + (X::EXACT || j < n ? j : n - 1);
            vx[t] = load_sample(x, idx, CPLX);
            vy[t] = load_sample(y, yoff + idx, CPLX);
        }
        if (segmean) {
            segment_detrend<C>(vx, lds, tid, n, X::EXACT, segmean);
            segment_detrend<C>(vy, lds, tid, n, X::EXACT, segmean);
        }
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int64_t idx = base + tid + C::T * t;
            vx[t] = w[t] * detrended<LIN>(vx[t], trx, idx);
            vy[t] = w[t] * detrended<LIN>(vy[t], try_, idx);
        }
        fwd_row(xf, vx, lds, tid, n);
        fwd_row(xf, vy, lds, tid, n);
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            axx[t] += keep * cnorm(vx[t]);
            ayy[t] += keep * cnorm(vy[t]);
            cf p = cmulc(vy[t], vx[t]);     // Y conj(X)  (fft_analysis.py:393)
            axy[t] = axy[t] + keep * p;
        }
    }
    float *p = partial + ((int64_t)ch * groups_total + gid) * 4 * X::L;
#pragma unroll
    for (int t = 0; t < C::R; ++t) {
        const int k = tid + C::T * t;
        p[k] = axx[t];
        p[X::L + k] = ayy[t];
        p[2 * X::L + k] = axy[t].x;
        p[3 * X::L + k] = axy[t].y;
    }
}

// Real x and y: z = w (x - mx) + i w (y_c - my) -> one transform per (frame, channel).  With Zm = Z[n-k]:
//   X = (Z + conj Zm)/2,  Y = (Z - conj Zm)/(2i)
//   |X|^2 = (|Z|^2 + |Zm|^2 + 2 Re(Z Zm))/4,  |Y|^2 = (|Z|^2 + |Zm|^2 - 2 Re(Z Zm))/4,
//   Y conj(X) = Im(Z Zm)/2 - i (|Z|^2 - |Zm|^2)/4
// so the kernel accumulates a[k] = |Z[k]|^2 and c[k] = Z[k] Zm[k] (mirror through one LDS exchange) and the finish
// kernel does the algebra.  partial per (channel, group): [3][L] = a, Re c, Im c.  Power-of-two n.
template <int N, bool LIN>
__global__ __launch_bounds__(WgCfg<N>::WG) void k_welch_csd_rp(const float *__restrict__ x, const float *__restrict__ y,
                                                                int64_t y_ld, const float *__restrict__ win, int hop,
                                                                int64_t nframes, int64_t fpg,
                                                                const float *__restrict__ trend_x,
                                                                const float *__restrict__ trend_y, XfTables tb,
                                                                float *__restrict__ partial, int64_t groups_total) {
    using X = XfPow2<N>;
    SP_KERNEL_PROLOGUE(X)
    (void)n;
    const int ch = blockIdx.y;
    float w[C::R], aa[C::R];
    cf cc[C::R];
#pragma unroll
    for (int t = 0; t < C::R; ++t) {
        w[t] = win[tid + C::T * t];
        aa[t] = 0.f;
        cc[t] = mk(0.f, 0.f);
    }
    const Trend trx = load_trend(trend_x), try_ = load_trend(trend_y + 4 * ch);
    const float *yc = y + (int64_t)ch * y_ld;
    const int64_t gid = (int64_t)blockIdx.x * C::FPW + grp;
    const int64_t g0 = gid * fpg;
    for (int64_t i = 0; i < fpg; ++i) {
        const int64_t g = g0 + i;
        const float keep = g < nframes ? 1.f : 0.f;
        const int64_t base = (g < nframes ? g : nframes - 1) * hop;
        cf v[C::R];
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int64_t idx = base + tid + C::T * t;
            v[t] = mk(x[idx], yc[idx]);
        }
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int64_t idx = base + tid + C::T * t;
            const cf a = detrended<LIN>(mk(v[t].x, 0.f), trx, idx);
            const cf b = detrended<LIN>(mk(v[t].y, 0.f), try_, idx);
            v[t] = mk(w[t] * a.x, w[t] * b.x);
        }
        fwd_row(xf, v, lds, tid, N);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < C::R; ++t) lds[tid + C::T * t] = v[t];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int k = tid + C::T * t;
            const cf zm = lds[(N - k) & (N - 1)];
            aa[t] += keep * cnorm(v[t]);
            cc[t] = cc[t] + keep * cmul(v[t], zm);
        }
    }
    float *p = partial + ((int64_t)ch * groups_total + gid) * 3 * N;
#pragma unroll
    for (int t = 0; t < C::R; ++t) {
        const int k = tid + C::T * t;
        p[k] = aa[t];
        p[N + k] = cc[t].x;
        p[2 * N + k] = cc[t].y;
    }
}

// ---- reference x against MANY real channels: the reference is transformed once ------------------------------------
// (fft_analysis.py:387-393 loops the channels against one x)  Every real signal packs two consecutive
// frames into one transform, Z = F_g + i F_{g+1}.  With A[k] = sum_pairs Zy[k] conj(Zx[k]), the real-input symmetries
// give   sum_g Y_g conj(X_g) = (A[k] + conj(A[n-k])) / 2   and   sum_g |Y_g|^2 = (a[k] + a[n-k]) / 2,  a = sum |Zy|^2
// -- the cross-frame terms cancel in the mirror combination, which is taken once, on the accumulated sums, by the finish
// kernel.  So: k_pairspec writes the reference's packed pair spectra Zx once (npairs x n complex, L2/MALL resident),
// k_welch_csd_pair does ONE transform per (channel, frame pair) -- half of the x + i y_c form, and without its per-frame
// mirror exchange -- and reads Zx.  partial per (channel, group): [3][L] = a, Re A, Im A.  Power-of-two n.
template <int N, bool LIN>
__global__ __launch_bounds__(WgCfg<N>::WG) void k_pairspec(const float *__restrict__ x, const float *__restrict__ win, int hop,
                                                            int64_t nframes, int64_t ppg, const float *__restrict__ trend,
                                                            XfTables tb, cf *__restrict__ out) {
    using X = XfPow2<N>;
    SP_KERNEL_PROLOGUE(X)
    (void)n;
    float w[C::R];
#pragma unroll
    for (int t = 0; t < C::R; ++t) w[t] = win[tid + C::T * t];
    const Trend tr = load_trend(trend);
    const int64_t npairs = (nframes + 1) / 2;
    const int64_t gid = (int64_t)blockIdx.x * C::FPW + grp;
    const int64_t p0 = gid * ppg;
    for (int64_t i = 0; i < ppg; ++i) {
        const int64_t p = p0 + i;
        const int64_t ga = 2 * (p < npairs ? p : npairs - 1);
        const bool has_b = ga + 1 < nframes;
        const int64_t base_a = ga * hop, base_b = (has_b ? ga + 1 : ga) * hop;
        const float kb = has_b ? 1.f : 0.f;
        cf v[C::R];
#pragma unroll
        for (int t = 0; t < C::R; ++t) v[t] = mk(x[base_a + tid + C::T * t], x[base_b + tid + C::T * t]);
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int j = tid + C::T * t;
            const cf a = detrended<LIN>(mk(v[t].x, 0.f), tr, base_a + j);
            const cf b = detrended<LIN>(mk(v[t].y, 0.f), tr, base_b + j);
            v[t] = mk(w[t] * a.x, kb * w[t] * b.x);
        }
        xf.fwd(v, lds, tid, N);
        if (p < npairs) {
#pragma unroll
            for (int t = 0; t < C::R; ++t) out[p * N + tid + C::T * t] = v[t];
        }
    }
}

// OP (mean detrend in one pass, hop = N/2, one transform per workgroup): every workgroup of a channel forms the same estimate
// mu0 of the channel's mean (16 runs of WG samples), detrends by it and leaves the block sums of its frames' last hop-blocks in
// spartial[ch][group][hop]; the finish kernel corrects with the exact mean (k_csd_pair_finish).  Workgroup 0 publishes mu0.
template <int N, bool LIN, bool OP = false>
__global__ __launch_bounds__(WgCfg<N>::WG) __attribute__((amdgpu_waves_per_eu(OP ? 2 : 1, OP ? 2 : 8))) void k_welch_csd_pair(const float *__restrict__ y, int64_t y_ld,
                                                                  const float *__restrict__ win, int hop, int64_t nframes,
                                                                  int64_t ppg, float *__restrict__ trend_y, XfTables tb,
                                                                  const cf *__restrict__ Zx, float *__restrict__ partial,
                                                                  int64_t groups_total, int chan_fast, cf *__restrict__ spartial) {
    using X = XfPow2<N>;
    SP_KERNEL_PROLOGUE(X)
    (void)n;
    static_assert(!OP || (!LIN && WgCfg<N>::FPW == 1 && (C::R % 2) == 0), "one-pass form: mean detrend, one transform per workgroup");
    // chan_fast: the channel is the fastest-varying block index, so that the workgroups resident at any time are the same few
    // runs of pairs of ALL channels and share the reference's pair spectra Zx through the L2s (with the run fastest every
    // channel streamed its own copy of the 134 MB from the Infinity Cache: 8.4 GB at 63 channels)
    const int ch = chan_fast ? blockIdx.x : blockIdx.y;
    const int bx = chan_fast ? blockIdx.y : blockIdx.x;
    float w[C::R], aa[C::R];
    cf cc[C::R];
#pragma unroll
    for (int t = 0; t < C::R; ++t) {
        w[t] = win[tid + C::T * t];
        aa[t] = 0.f;
        cc[t] = mk(0.f, 0.f);
    }
    const float *yc = y + (int64_t)ch * y_ld;
    Trend tr;
    float sacc[OP ? C::R / 2 : 1];
    if constexpr (OP) {
        const int64_t span = (nframes - 1) * (int64_t)hop + N;
        const int64_t pitch = span / 16;
        float sx = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int64_t i = pitch * r + (int64_t)threadIdx.x;
            i = i < span ? i : span - 1;
            sx += yc[i];
        }
        sx = wave_sum64(sx);
        float *red = reinterpret_cast<float *>(smem);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sx;
        __syncthreads();
        double tx = 0.0;
#pragma unroll
        for (int wv = 0; wv < C::WG / 64; ++wv) tx += (double)red[wv];
        __syncthreads();
        tr.m = mk((float)(tx / (16.0 * C::WG)), 0.f);
        tr.s = mk(0.f, 0.f);
        if (bx == 0 && threadIdx.x == 0) {
            trend_y[4 * ch] = tr.m.x;
            trend_y[4 * ch + 1] = 0.f;
            trend_y[4 * ch + 2] = 0.f;
            trend_y[4 * ch + 3] = 0.f;
        }
#pragma unroll
        for (int s2 = 0; s2 < C::R / 2; ++s2) sacc[s2] = 0.f;
    } else {
        tr = load_trend(trend_y + 4 * ch);
    }
    const int64_t npairs = (nframes + 1) / 2;
    const int64_t gid = (int64_t)bx * C::FPW + grp;
    const int64_t p0 = gid * ppg;
    // pair p -> its two frame bases (clamped past the end; a lone last frame has no second member)
    auto bases = [&](int64_t p, int64_t &base_a, int64_t &base_b, float &kb) __attribute__((always_inline)) {
        const int64_t pc = p < npairs ? p : npairs - 1;
        const int64_t ga = 2 * pc;
        const bool has_b = ga + 1 < nframes;
        base_a = ga * hop;
        base_b = (has_b ? ga + 1 : ga) * hop;
        kb = has_b ? 1.f : 0.f;
    };
    cf raw[C::R];
    {
        int64_t ba, bb;
        float kb;
        bases(p0, ba, bb, kb);
#pragma unroll
        for (int t = 0; t < C::R; ++t) raw[t] = mk(yc[ba + tid + C::T * t], yc[bb + tid + C::T * t]);
    }
    for (int64_t i = 0; i < ppg; ++i) {
        const int64_t p = p0 + i;
        const float keep = p < npairs ? 1.f : 0.f;
        const int64_t pc = p < npairs ? p : npairs - 1;
        int64_t base_a, base_b;
        float kb;
        bases(p, base_a, base_b, kb);
        cf v[C::R], zx[C::R];
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int j = tid + C::T * t;
            const cf a = detrended<LIN>(mk(raw[t].x, 0.f), tr, base_a + j);
            const cf b = detrended<LIN>(mk(raw[t].y, 0.f), tr, base_b + j);
            v[t] = mk(w[t] * a.x, kb * w[t] * b.x);
            if constexpr (OP) {
                if (t >= C::R / 2) sacc[t - C::R / 2] += keep * (a.x + kb * b.x);       // last hop-block of both frames
            }
        }
        // in flight during the transform: the samples of the next pair (the reference's spectrum of this pair is read
        // after it: with both in flight the kernel needs 261 VGPRs = one wave per SIMD)
        {
            int64_t na, nb2;
            float kn;
            bases(p + 1, na, nb2, kn);
#pragma unroll
            for (int t = 0; t < C::R; ++t) raw[t] = mk(yc[na + tid + C::T * t], yc[nb2 + tid + C::T * t]);
        }
        xf.fwd(v, lds, tid, N);
#pragma unroll
        for (int t = 0; t < C::R; ++t) zx[t] = Zx[pc * N + tid + C::T * t];
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            aa[t] += keep * cnorm(v[t]);
            cc[t] = cc[t] + keep * cmulc(v[t], zx[t]);
        }
    }
    float *p = partial + ((int64_t)ch * groups_total + gid) * 3 * N;
#pragma unroll
    for (int t = 0; t < C::R; ++t) {
        const int k = tid + C::T * t;
        p[k] = aa[t];
        p[N + k] = cc[t].x;
        p[2 * N + k] = cc[t].y;
    }
    if constexpr (OP) {
#pragma unroll
        for (int s2 = 0; s2 < C::R / 2; ++s2)
            spartial[((int64_t)ch * groups_total + gid) * hop + tid + C::T * s2] = mk(sacc[s2], 0.f);
    }
}

// block sums of ONE real signal (the reference of the one-pass pair path): slice blockIdx.y of the hop-blocks b = 1 .. M writes
// out[slice][j] = sum_b (x[b H + j] - mu); k_cm_blocksums adds the slices in a fixed order (no atomics: reproducible)
static __global__ void k_colsum_real(const float *__restrict__ x, const float *__restrict__ trend, int H, int64_t M, cf *__restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= H) return;
    const float mu = trend[0];
    const int64_t per = (M + gridDim.y - 1) / gridDim.y;
    const int64_t b0 = 1 + (int64_t)blockIdx.y * per, b1 = b0 + per < M + 1 ? b0 + per : M + 1;
    double a = 0.0;
#pragma unroll 8
    for (int64_t b = b0; b < b1; ++b) a += (double)(x[b * H + j] - mu);
    out[(int64_t)blockIdx.y * H + j] = mk((float)a, 0.f);
}

// pyy[ch][slot] = (a[k] + a[n-k]) / 2,  pxy[ch][slot] = (A[k] + conj(A[n-k])) / 2, scaled / doubled per sidedness
// st_y != null (one-pass mean detrend): the channels were detrended by estimates mu0; with d = mean - mu0 (real), W = FFT(window),
// B = sum_g of the spectra (k_op_finish<EXPORT> states st_y[ch], st_x), M frames, nmean samples per signal:
//   sum |Y - dy W|^2 = a - 2 Re(conj(dy W) By) + M |dy W|^2,   sum (Y - dy W) conj(X - dx W) = A - dx conj(W) By - dy W conj(Bx) + M dx dy |W|^2
static __global__ __launch_bounds__(SP_FIN_BINS *SP_FIN_SLICES) void k_csd_pair_finish(const float *__restrict__ partial,
                                                                                 int64_t G, int n, int nch, int sided,
                                                                                 double scale, double *__restrict__ pyy,
                                                                                 double *__restrict__ pxy,
                                                                                 const double *__restrict__ st_y,
                                                                                 const double *__restrict__ st_x,
                                                                                 const cf *__restrict__ Wf,
                                                                                 const float *__restrict__ trend_x,
                                                                                 const float *__restrict__ trend_y, int64_t nmean,
                                                                                 int64_t M) {
    __shared__ double sh[6][SP_FIN_SLICES][SP_FIN_BINS];
    const int lane = threadIdx.x % SP_FIN_BINS, sl = threadIdx.x / SP_FIN_BINS;
    const int k = blockIdx.x * SP_FIN_BINS + lane;
    const int ch = blockIdx.y;
    const int nb = nbins_of(n, sided);
    double s[6] = {0, 0, 0, 0, 0, 0};       // a[k], a[km], Re A[k], Im A[k], Re A[km], Im A[km]
    const float *p = partial + (int64_t)ch * G * 3 * n;
    if (k < n) {
        const int km = k == 0 ? 0 : n - k;
        for (int64_t g = sl; g < G; g += SP_FIN_SLICES) {
            s[0] += (double)p[(g * 3 + 0) * n + k];
            s[1] += (double)p[(g * 3 + 0) * n + km];
            s[2] += (double)p[(g * 3 + 1) * n + k];
            s[3] += (double)p[(g * 3 + 2) * n + k];
            s[4] += (double)p[(g * 3 + 1) * n + km];
            s[5] += (double)p[(g * 3 + 2) * n + km];
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) sh[j][sl][lane] = s[j];
    fin_reduce<6>(sh, sl, lane);
    if (sl == 0 && k < n) {
        const int slot = bin_slot(k, n, sided);
        if (slot >= 0) {
            double t[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) t[j] = sh[j][0][lane];
            const double m = 0.5 * scale * (bin_doubled(k, n, sided) ? 2.0 : 1.0);
            double cyy = 0.0, cr = 0.0, ci = 0.0;
            if (st_y) {
                const int64_t ss = (int64_t)5 * n + 8;
                const double *sy = st_y + ch * ss;
                const double dy = sy[5 * n + 3] / (double)nmean - (double)trend_y[4 * ch];
                const double dx = st_x[5 * n + 3] / (double)nmean - (double)trend_x[0];
                const double wr = Wf[k].x, wi = Wf[k].y, w2 = wr * wr + wi * wi;
                const double byr = sy[n + 2 * k], byi = sy[n + 2 * k + 1], bxr = st_x[n + 2 * k], bxi = st_x[n + 2 * k + 1];
                cyy = -2.0 * dy * (wr * byr + wi * byi) + (double)M * dy * dy * w2;
                cr = -dx * (wr * byr + wi * byi) - dy * (wr * bxr + wi * bxi) + (double)M * dx * dy * w2;
                ci = -dx * (wr * byi - wi * byr) - dy * (wi * bxr - wr * bxi);
            }
            pyy[(int64_t)ch * nb + slot] = (t[0] + t[1] + 2.0 * cyy) * m;
            pxy[((int64_t)ch * nb + slot) * 2] = (t[2] + t[4] + 2.0 * cr) * m;
            pxy[((int64_t)ch * nb + slot) * 2 + 1] = (t[3] - t[5] + 2.0 * ci) * m;
        }
    }
}

static __global__ __launch_bounds__(SP_FIN_BINS *SP_FIN_SLICES) void k_csd_rp_finish(const float *__restrict__ partial,
                                                                               int64_t G, int n, int nch, int sided,
                                                                               double scale, double *__restrict__ pxx,
                                                                               double *__restrict__ pyy,
                                                                               double *__restrict__ pxy) {
    __shared__ double sh[4][SP_FIN_SLICES][SP_FIN_BINS];
    const int lane = threadIdx.x % SP_FIN_BINS, sl = threadIdx.x / SP_FIN_BINS;
    const int k = blockIdx.x * SP_FIN_BINS + lane;
    const int ch = blockIdx.y;
    const int nb = nbins_of(n, sided);
    double s[4] = {0, 0, 0, 0};       // a[k], a[km], Re c[k], Im c[k]
    const float *p = partial + (int64_t)ch * G * 3 * n;
    if (k < n) {
        const int km = k == 0 ? 0 : n - k;
        for (int64_t g = sl; g < G; g += SP_FIN_SLICES) {
            s[0] += (double)p[(g * 3 + 0) * n + k];
            s[1] += (double)p[(g * 3 + 0) * n + km];
            s[2] += (double)p[(g * 3 + 1) * n + k];
            s[3] += (double)p[(g * 3 + 2) * n + k];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) sh[j][sl][lane] = s[j];
    fin_reduce<4>(sh, sl, lane);
    if (sl == 0 && k < n) {
        const int slot = bin_slot(k, n, sided);
        if (slot >= 0) {
            double t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = sh[j][0][lane];
            const double m = scale * (bin_doubled(k, n, sided) ? 2.0 : 1.0);
            if (ch == 0) pxx[slot] = 0.25 * (t[0] + t[1] + 2.0 * t[2]) * m;
            pyy[(int64_t)ch * nb + slot] = 0.25 * (t[0] + t[1] - 2.0 * t[2]) * m;
            pxy[((int64_t)ch * nb + slot) * 2] = 0.5 * t[3] * m;
            pxy[((int64_t)ch * nb + slot) * 2 + 1] = -0.25 * (t[0] - t[1]) * m;
        }
    }
}

// out layouts: pxx[nbins] (from channel 0's copy), pyy[nch][nbins], pxy[nch][nbins][2]
static __global__ __launch_bounds__(SP_FIN_BINS *SP_FIN_SLICES) void k_csd_finish(const float *__restrict__ partial, int64_t G,
                                                                            int L, int n, int nch, int sided,
                                                                            double scale, double *__restrict__ pxx,
                                                                            double *__restrict__ pyy,
                                                                            double *__restrict__ pxy) {
    __shared__ double sh[4][SP_FIN_SLICES][SP_FIN_BINS];
    const int lane = threadIdx.x % SP_FIN_BINS, sl = threadIdx.x / SP_FIN_BINS;
    const int k = blockIdx.x * SP_FIN_BINS + lane;
    const int ch = blockIdx.y;
    const int nb = nbins_of(n, sided);
    double s[4] = {0, 0, 0, 0};
    const float *p = partial + (int64_t)ch * G * 4 * L;
    if (k < n)
        for (int64_t g = sl; g < G; g += SP_FIN_SLICES) {
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += (double)p[(g * 4 + j) * L + k];
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) sh[j][sl][lane] = s[j];
    fin_reduce<4>(sh, sl, lane);
    if (sl == 0 && k < n) {
        const int slot = bin_slot(k, n, sided);
        if (slot >= 0) {
            double tot[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) tot[j] = sh[j][0][lane];
            const double m = scale * (bin_doubled(k, n, sided) ? 2.0 : 1.0);
            if (ch == 0) pxx[slot] = tot[0] * m;
            pyy[(int64_t)ch * nb + slot] = tot[1] * m;
            pxy[((int64_t)ch * nb + slot) * 2] = tot[2] * m;
            pxy[((int64_t)ch * nb + slot) * 2 + 1] = tot[3] * m;
        }
    }
}

// ------------------------------------------------------------------------------------------
// A8/A9  STFT frames (fft_analysis.py:2156-2203; spectrogram.py:91-112).
// out frame-major [nframes][nbins]; complex (amp * X, sqrt2 on doubled bins) or power (amp*|X|^2).
// pseg (optional): trapz of |win*(x-trend)|^2 over the frame, unit spacing (:2174).
// ------------------------------------------------------------------------------------------
template <class X, bool CPLX, bool LIN>
__global__ __launch_bounds__(X::C::WG) void k_stft(const void *__restrict__ x, const float *__restrict__ win, int hop,
                                                    int64_t nframes, int64_t fpg, const float *__restrict__ trend,
                                                    XfTables tb, int sided, float amp, int out_power,
                                                    void *__restrict__ out, double *__restrict__ pseg, int segmean,
                                                    cf *__restrict__ cog, int klo, int khi) {
    SP_KERNEL_PROLOGUE(X)
    float w[C::R];
#pragma unroll
    for (int t = 0; t < C::R; ++t) {
        const int i = tid + C::T * t;
        w[t] = (X::EXACT || i < n) ? win[i] : 0.f;
    }
    const Trend tr = load_trend(trend);
    const int nb = nbins_of(n, sided);
    const int64_t gid = (int64_t)blockIdx.x * C::FPW + grp;
    const int64_t g0 = gid * fpg;
    for (int64_t i = 0; i < fpg; ++i) {
        const int64_t g = g0 + i;
        const bool act = g < nframes;
        const int64_t base = (act ? g : nframes - 1) * hop;
        cf v[C::R];
        float pw = 0.f;
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int j = tid + C::T * t;
            v[t] = load_sample(x, base + (X::EXACT || j < n ? j : n - 1), CPLX);
        }
        if (segmean) segment_detrend<C>(v, lds, tid, n, X::EXACT, segmean);   // per-window detrend (fft_win detrendwin=True)
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int j = tid + C::T * t;
            v[t] = w[t] * detrended<LIN>(v[t], tr, base + j);
            pw += ((j == 0 || j == n - 1) ? 0.5f : 1.f) * cnorm(v[t]);
        }
        if (pseg != nullptr) {        // one atomic per wave and frame instead of one per thread
            constexpr int W = C::T < 64 ? C::T : 64;
            pw = group_lane_sum<W>(pw);
            if (act && (tid & (W - 1)) == 0) atomicAdd(&pseg[g], (double)pw);
        }
        fwd_row(xf, v, lds, tid, n);
        if (cog != nullptr) {
            // centre of gravity of the frame's two-sided power spectrum (Doppler.py:43-58): moments sum |X|^2 ks and
            // sum |X|^2 over the signed bin index ks = fftfreq(n) n, band klo <= |ks| <= khi; the caller divides
            float num = 0.f, den = 0.f;
#pragma unroll
            for (int t = 0; t < C::R; ++t) {
                const int k = tid + C::T * t;
                if (!X::EXACT && k >= n) continue;
                const int ks = k < (n + 1) / 2 ? k : k - n;
                const int ka = ks < 0 ? -ks : ks;
                const float p = (ka >= klo && ka <= khi) ? cnorm(v[t]) : 0.f;
                num += p * (float)ks;
                den += p;
            }
            constexpr int W = C::T < 64 ? C::T : 64;       // lanes of one frame group inside a wave
#pragma unroll
            for (int o = W / 2; o > 0; o >>= 1) {
                num += __shfl_xor(num, o);
                den += __shfl_xor(den, o);
            }
            if ((tid & (W - 1)) == 0 && act) {
                cog[(int64_t)(tid / W) * nframes + g] = mk(num, den);   // slot [wave of the frame][frame], summed by k_cog_finish
            }
            continue;
        }
        if (act) {
#pragma unroll
            for (int t = 0; t < C::R; ++t) {
                const int k = tid + C::T * t;
                if (!X::EXACT && k >= n) continue;
                const int slot = bin_slot(k, n, sided);
                if (slot < 0) continue;
                if (out_power) {
                    reinterpret_cast<float *>(out)[g * nb + slot] = amp * cnorm(v[t]);
                } else {
                    const float a = bin_doubled(k, n, sided) ? amp * 1.41421356237309504880f : amp;
                    reinterpret_cast<cf *>(out)[g * nb + slot] = a * v[t];
                }
            }
        }
    }
}

// cog[g] = df * num / den (0 where the band holds no power) from the per-wave moment slots acc[wpf][g] = (num, den)
static __global__ void k_cog_finish(const cf *__restrict__ acc, int wpf, int64_t nframes, double df, double *__restrict__ out) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nframes) return;
    double num = 0.0, den = 0.0;
    for (int w = 0; w < wpf; ++w) {
        const cf a = acc[(int64_t)w * nframes + g];
        num += (double)a.x;
        den += (double)a.y;
    }
    out[g] = den > 0.0 ? df * num / den : 0.0;
}

// bins ks = -K .. K of FFT(window) (float64, from the host), K <= 3: the lobe of a cosine-sum window
struct CogLobe {
    int K;
    double wr[7], wi[7];
};
// the same with the one-pass mean detrend of a cosine-sum window (k_welch_pipe mode 8): the spectra were detrended by the estimate
// mu0; with d = mean - mu0 (st: state of k_op_finish<EXPORT>, plain sample sums at 5n + 3, 5n + 4) and W = FFT(window), non-zero
// in the bins ks = -K .. K only, |X - d W|^2 - |X|^2 = -2 Re(conj(d W) X) + |d W|^2 there: 2K + 1 terms per frame from lobe[g][ks + 3]
static __global__ void k_cog_finish_op(const cf *__restrict__ acc, int wpf, int64_t nframes, double df, double *__restrict__ out,
                                       const cf *__restrict__ lobe, CogLobe lb, const double *__restrict__ st,
                                       const float *__restrict__ trend, int64_t nmean, int n) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nframes) return;
    double num = 0.0, den = 0.0;
    for (int w = 0; w < wpf; ++w) {
        const cf a = acc[(int64_t)w * nframes + g];
        num += (double)a.x;
        den += (double)a.y;
    }
    const double dr = st[5 * n + 3] / (double)nmean - (double)trend[0], di = st[5 * n + 4] / (double)nmean - (double)trend[1];
    for (int ks = -lb.K; ks <= lb.K; ++ks) {
        const double wr = lb.wr[ks + 3], wi = lb.wi[ks + 3];
        const double er = dr * wr - di * wi, ei = dr * wi + di * wr;             // d W
        const cf x = lobe[g * 8 + ks + 3];
        const double dp = -2.0 * (er * (double)x.x + ei * (double)x.y) + er * er + ei * ei;
        den += dp;
        num += (double)ks * dp;
    }
    out[g] = den > 0.0 ? df * num / den : 0.0;
}

// ---- the one-pass Welch epilogue in ONE launch, for windows whose spectrum is confined to the bins -3 .. 3 -------------
// k_op_colsums + k_op_finish are two launches, the second one a single workgroup running a chain of memory round trips and a
// transform: 7 + 19 us of kernels and two boundaries per step, a third of a 2^25-sample shard's step (strong scaling at 8
// GPUs).  The transform B = FFT(w c) = sum_g X_g is only needed where W = FFT(window) is non-zero, P[k] = A[k] -
// 2 Re(conj(d W[k]) B[k]) + M |d W[k]|^2: for the periodic cosine-sum windows (Hann, Hamming, Blackman ...) these are the
// bins ks = -K .. K, K <= 3 (CogLobe, float64 from the host, checked by Parseval) -- 2K + 1 direct sums over n instead of a
// transform.  So: a column-sum grid (16-byte loads, every load of a thread in flight at once), the sums published with
// write-through (sc1) stores and a ticket per block, and the block that arrives last (no spinning: nothing waits for another
// block) finishes from sc1 loads -- the hand-off form of MI355X_MICROARCH.md "Valid forms": sc1 stores drained by the storing
// wave, ONE lane's agent-scope atomic add, the last adder's block loads behind a workgroup barrier; no release / acquire
// fence (1.7-6.5 us each).  The last block issues everything it needs in ONE round trip (block sums, raw sums, window, the
// edge blocks of c[n]), reduces the mean's total and the 2K + 1 lobe sums in ONE block reduction, and writes the spectrum or
// the additive shard state; its twiddles e^{-2 pi i n / N} are computed before the ticket, off the critical path.
// `ticket` is zero on entry and is left zero.
// EXPORT: out = the state of k_op_finish<EXPORT> (A | B | conj(mu0) B | scalars) with B zero outside the lobe bins, which is
// all k_op_apply multiplies by a non-zero W.  prev != null (EXPORT): the last block also turns the all-reduced state of the
// PREVIOUS step into that step's spectrum (k_op_apply's arithmetic; pipelined sharded PSD, sp_welch_dist_submit).
__device__ __forceinline__ double wave_sum64d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// SP_OPF_VARIANT (diagnostic builds): bit 0: the ticket is reset with a plain store; bit 1: plain stores / loads with agent-scope
// release / acquire fences instead of the sc1 forms
#ifndef SP_OPF_VARIANT
#define SP_OPF_VARIANT 1
#endif
__device__ __forceinline__ void st_sc1(double *p, double v) {
    if constexpr (SP_OPF_VARIANT & 2) *p = v;
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double *p) {
    if constexpr (SP_OPF_VARIANT & 2) return *p;
    else return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#if SP_OPF_VARIANT & 4
__device__ unsigned long long g_opf_t0[1024], g_opf_t1[1024];
__device__ unsigned g_opf_launches;
#define OPF_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memrealtime()
#else
#define OPF_STAMP(v)
#endif
struct OpPrev {                 // the previous step of the pipelined sharded PSD (all pointers null: nothing to apply)
    const double *st;           // its all-reduced state [5 N + 8]
    const cf *Wf;               // FFT(window) (float), N bins
    double *out;                // its spectrum
    int sided;
    double scale;
};
// E = edge blocks of c[n] per side: 1 for hop = N/2 and hop = N, 3 for hop = N/4.  512 threads (256 VGPRs: the last block keeps
// eight bins per thread in flight; at 1024 threads / 128 VGPRs the same code spilled 450 registers and ran 5x slower).
// LIGHT: the form the streaming engine launches on its own stream so that it runs BESIDE the next step's main kernel: that kernel
// (k_welch_pipe: 3 waves of 136 VGPRs per SIMD, 128.5 of 160 KiB LDS) leaves 104 VGPRs per SIMD and plenty of wave slots, so a
// workgroup of 256 threads (one wave per SIMD) with <= 104 VGPRs fits next to it (the 512-thread form with 190 does not: its
// workgroups waited for the main kernel to drain and the overlap was lost, profiles/r03_stream_overlap.txt).  Two bins per thread
// and chunk: more round trips for the last block, which nobody waits for.  (Since the rotations moved one role upstream the
// main kernel allocates 144 VGPRs per wave, 80 are left per SIMD: the light form is held to 80 -- waves_per_eu 6 -- and spills
// a few registers in its last block, whose latency nobody sees.)
#define SP_OPF_WG 512
#define SP_OPF_WG_LIGHT 256
// LOBEB: the main kernel ran in mode 9 (k_welch_pipe): no block sums -- m1 is lobeB[G][8], the groups' sums of the spectra at the
// bins ks = -3 .. 3 (index ks + 3), and the window adds up to the constant cola_c at this hop.  B[ks] = sum_G lobeB; the plain sum
// of the samples follows from the DC bin: B[0] = sum_i cov(i) (x[i] - mu0), cov(i) = sum of the window values of the frames that
// cover sample i = cola_c everywhere but within N - H samples of the two ends, so sum_i (x[i] - mu0) = (B[0] + sum_edges (cola_c -
// cov(i)) (x[i] - mu0)) / cola_c -- a few thousand samples read by the last block.  No column sums of block sums (half of phase 1),
// no c[n] rebuild (the last block's big round trip).
template <bool CPLX, bool EXPORT, int E, bool LIGHT = false, bool LOBEB = false>
static __global__ __launch_bounds__(LIGHT ? SP_OPF_WG_LIGHT : SP_OPF_WG)
    __attribute__((amdgpu_waves_per_eu(LIGHT ? 6 : 2, LIGHT ? 8 : 2))) void k_op_fused(const float *__restrict__ m0, int N, double *__restrict__ Acol,
                                                                const float *__restrict__ m1, int H, double *__restrict__ Sl,
                                                                int64_t G, unsigned *__restrict__ ticket,
                                                                const void *__restrict__ x, const float *__restrict__ trend,
                                                                const float *__restrict__ win, CogLobe lb,
                                                                const double *__restrict__ mean_in, int64_t M, int64_t nmean,
                                                                int sided, double scale, double *__restrict__ out, int sym,
                                                                OpPrev prev, double step_c, double step_s, double cola_c) {
    constexpr int WG = LIGHT ? SP_OPF_WG_LIGHT : SP_OPF_WG, NW = WG / 64, NSL = WG / 8;
    OPF_STAMP(ts_start);
    __shared__ double sh[NW][32];
    __shared__ double tot_sh[16];
    __shared__ int last_flag;
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    {   // ---- phase 1: column sums of m0 [G][N] -> Acol and of m1 [G][2H] -> Sl.  A block owns 32 columns = 8 lanes of float4;
        // its NSL row slices: 8 per wave (summed by shuffles), NW waves (summed through LDS); 4 loads per thread in flight
        const int c0 = N, c1 = 2 * H;
        const int nb0 = c0 / 32;
        const bool second = (int)blockIdx.x >= nb0;
        const float *__restrict__ m = second ? m1 : m0;
        const int cols = second ? c1 : c0;
        double *__restrict__ o = second ? Sl : Acol;
        const int l8 = threadIdx.x & 7, sl = threadIdx.x >> 3;
        const int kb = ((int)blockIdx.x - (second ? nb0 : 0)) * 32;
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for (int64_t g0 = sl; g0 < G; g0 += 4 * NSL) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t g = g0 + NSL * u;
                v[u] = g < G ? *reinterpret_cast<const float4 *>(m + g * cols + kb + 4 * l8) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s[0] += (double)v[u].x;
                s[1] += (double)v[u].y;
                s[2] += (double)v[u].z;
                s[3] += (double)v[u].w;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s[q] += __shfl_xor(s[q], 8);
            s[q] += __shfl_xor(s[q], 16);
            s[q] += __shfl_xor(s[q], 32);
        }
        if (ln < 8) {
#pragma unroll
            for (int q = 0; q < 4; ++q) sh[wv][4 * ln + q] = s[q];
        }
        __syncthreads();
        if (threadIdx.x < 32) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += sh[w][threadIdx.x];
            st_sc1(o + kb + threadIdx.x, t);
        }
    }
    // ---- hand-off: the sc1 stores above all come from wave 0; it drains them, its lane 0 takes a ticket
    if (threadIdx.x < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if SP_OPF_VARIANT & 4
    OPF_STAMP(ts_p1);
    if (threadIdx.x == 0) {
        __hip_atomic_store(&g_opf_t0[blockIdx.x], ts_start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&g_opf_t1[blockIdx.x], ts_p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#endif
    if (threadIdx.x == 0) {
        if constexpr (SP_OPF_VARIANT & 2) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // two levels (256 adds on one word take 3 us): the blocks with equal blockIdx % 8 share a counter (64 bytes apart), the
        // last arriver of each group adds to the top counter, the last of those is the last block of the grid
        const unsigned grp = blockIdx.x & 7u, ngrp = gridDim.x < 8u ? gridDim.x : 8u;
        const unsigned members = (gridDim.x - 1u - grp) / 8u + 1u;
        int last = 0;
        if (__hip_atomic_fetch_add(ticket + 16 * grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1u)
            last = __hip_atomic_fetch_add(ticket + 16 * 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngrp - 1u;
        last_flag = last;
        if constexpr (SP_OPF_VARIANT & 2) {
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    }
    __syncthreads();
    if (!last_flag) return;
    OPF_STAMP(ts_tick);
#if SP_OPF_VARIANT & 4
    unsigned long long ts_loaded = 0;
#endif
    // ---- phase 2 (the last block to arrive; Acol / Sl are read with sc1 loads only)
    const cf mu = mk(trend[0], trend[1]);
    const int hs = __builtin_ctz((unsigned)H);                 // H is a power of two on this path (launch_op_fused checks)
    const int r = N >> hs;
    const int64_t cov = (M + r - 1) * (int64_t)H;
    // 16 sums of ONE block reduction: [0,1] sum_{i < nmean} (x[i] - mu0), [2 + 2 q, 3 + 2 q] B[ks = q - 3]
    double acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0;
    if (EXPORT || !mean_in) {         // the ragged end (nothing for a whole signal, one hop for a shard)
        const int64_t lo = nmean > cov ? cov : nmean, hi = nmean > cov ? nmean : cov;
        const double sgn = nmean > cov ? 1.0 : -1.0;
        for (int64_t i = lo + threadIdx.x; i < hi; i += WG) {
            const cf v = load_sample(x, i, CPLX) - mu;
            acc[0] += sgn * v.x;
            acc[1] += sgn * v.y;
        }
    }
    constexpr int NPT = LIGHT ? 2 : ((E > 1 && CPLX) ? 4 : 8);    // bins per thread and chunk: N <= 4096 is one chunk, one round trip (hop = N/4, complex: two)
    double ak[NPT];                                 // raw sums A[k] of the thread's bins (kept for the output loop)
    if constexpr (LOBEB) {
        const cf *__restrict__ lobeB = reinterpret_cast<const cf *>(m1);
        for (int64_t g = threadIdx.x; g < G; g += WG) {
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                const cf v = lobeB[g * 8 + q];
                acc[2 + 2 * q] += (double)v.x;
                acc[3 + 2 * q] += (double)v.y;
            }
        }
        if (EXPORT || !mean_in) {
            // the two edges, where fewer than r frames cover a sample: (cola_c - cov(i)) (x[i] - mu0) / cola_c
            const int64_t nhead = (int64_t)(r - 1) * H, mh = M * (int64_t)H;
            const int64_t tail0 = mh > nhead ? mh : nhead;
            const double ic = 1.0 / cola_c;
            for (int side = 0; side < 2; ++side) {
                const int64_t i0 = side ? tail0 : 0, i1 = side ? cov : (nhead < cov ? nhead : cov);
                for (int64_t i = i0 + threadIdx.x; i < i1; i += WG) {
                    int64_t ghi = i >> hs;
                    ghi = ghi < M - 1 ? ghi : M - 1;
                    const int64_t glo = i >= N ? ((i - N) >> hs) + 1 : 0;
                    double cv = 0.0;
                    for (int64_t gg = glo; gg <= ghi; ++gg) cv += (double)win[i - (gg << hs)];
                    const cf v = load_sample(x, i, CPLX) - mu;
                    const double d = (cola_c - cv) * ic;
                    acc[0] += d * (double)v.x;
                    acc[1] += d * (double)v.y;
                }
            }
        }
    } else {
    // uniform bases + 32-bit lane offsets (scalar-base addressing: one offset register per load instead of a 64-bit address)
    const char *xh = reinterpret_cast<const char *>(x), *xt = xh + M * (int64_t)H * (CPLX ? 8 : 4);
    for (int base = 0; base < N; base += WG * NPT) {
        double slr[NPT], sli[NPT];
        float wn[NPT];
        cf eh[NPT][E], et[NPT][E];
#pragma unroll
        for (int t = 0; t < NPT; ++t) {
            const unsigned nidx = (unsigned)(base + (int)threadIdx.x + WG * t);
            const bool on = nidx < (unsigned)N;
            const unsigned nc = on ? nidx : 0u;
            const unsigned q = nc >> hs, j = nc & (unsigned)(H - 1);
            slr[t] = ld_sc1(Sl + 2u * j);
            sli[t] = ld_sc1(Sl + 2u * j + 1u);
            ak[t] = ld_sc1(Acol + nc);
            wn[t] = win[nc];
#pragma unroll
            for (int e = 0; e < E; ++e) {           // edge blocks of c[n]: b = q + e <= r - 2 (head), b = M + q + e <= M + r - 2 (tail)
                const bool he = on && (int)q + e <= r - 2;
                const unsigned off = he ? ((q + (unsigned)e) << hs) + j : 0u;
                const cf vh = load_sample(xh, off, CPLX), vt = load_sample(xt, off, CPLX);
                eh[t][e] = he ? vh - mu : mk(0.f, 0.f);
                et[t][e] = he ? vt - mu : mk(0.f, 0.f);
            }
        }
        __builtin_amdgcn_sched_barrier(0);          // (all loads issued above; consumed one bin at a time below)
#if SP_OPF_VARIANT & 4
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#if SP_OPF_VARIANT & 4
        ts_loaded = __builtin_amdgcn_s_memrealtime();
#endif
        // e^{-i theta_n}, theta_n = 2 pi n / N, for n = base + tid, then rotated by the step e^{-2 pi i WG / N} from bin to bin
        double c1, s1;
        sincospi(-2.0 * (double)(base + (int)threadIdx.x) / (double)N, &s1, &c1);
#pragma unroll
        for (int t = 0; t < NPT; ++t) {
            const int nidx = base + (int)threadIdx.x + WG * t;
            if (nidx < N) {
                const int q = nidx >> hs;
                double a = slr[t], b = sli[t];
                if (q == 0) {                     // every j once: the block sums, and below the head blocks i < (r - 1) H
                    acc[0] += a;
                    acc[1] += b;
                }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if (q == 0) {
                        acc[0] += (double)eh[t][e].x;
                        acc[1] += (double)eh[t][e].y;
                    }
                    a += (double)eh[t][e].x - (double)et[t][e].x;
                    b += (double)eh[t][e].y - (double)et[t][e].y;
                }
                a *= (double)wn[t];
                b *= (double)wn[t];
                acc[2 + 6] += a;
                acc[3 + 6] += b;
                double pr = 1.0, pi_ = 0.0;
#pragma unroll
                for (int ks = 1; ks <= 3; ++ks) {
                    const double tr_ = pr * c1 - pi_ * s1, ti_ = pr * s1 + pi_ * c1;      // e^{-i ks theta}
                    pr = tr_;
                    pi_ = ti_;
                    acc[2 + 2 * (3 + ks)] += a * pr - b * pi_;
                    acc[3 + 2 * (3 + ks)] += a * pi_ + b * pr;
                    acc[2 + 2 * (3 - ks)] += a * pr + b * pi_;                          // the conjugate phase for -ks
                    acc[3 + 2 * (3 - ks)] += b * pr - a * pi_;
                }
            }
            const double nc1 = c1 * step_c - s1 * step_s, ns1 = c1 * step_s + s1 * step_c;
            c1 = nc1;
            s1 = ns1;
        }
    }
    }
    OPF_STAMP(ts_math);
    // across the wave: the mean's total in double (shuffles), the lobe sums in float through DPP (no LDS round trips; they enter
    // the spectrum multiplied by d = mean - mu0: float32 is ample); across the waves: LDS
    acc[0] = wave_sum64d(acc[0]);
    acc[1] = wave_sum64d(acc[1]);
#pragma unroll
    for (int q = 2; q < 16; ++q) acc[q] = (double)wave_sum64((float)acc[q]);
    __syncthreads();                               // (phase 1's use of sh is over in every wave)
    if (ln == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) sh[wv][q] = acc[q];
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) a += sh[w][threadIdx.x];
        tot_sh[threadIdx.x] = a;
    }
    __syncthreads();
    // (LOBEB: the samples' sum = ragged end + edges / c [both in slot 0, 1] + B[0] / c)
    const double tot_r = tot_sh[0] + (LOBEB ? tot_sh[2 + 6] / cola_c : 0.0), tot_i = tot_sh[1] + (LOBEB ? tot_sh[3 + 6] / cola_c : 0.0);
    double dr, di;
    if (!EXPORT && mean_in) {
        dr = mean_in[0] - (double)trend[0];
        di = mean_in[1] - (double)trend[1];
    } else {
        dr = tot_r / (double)nmean;
        di = tot_i / (double)nmean;
    }
    OPF_STAMP(ts_red);
    const double mr = (double)mu.x, mi = (double)mu.y;
    const bool one_chunk = !LOBEB && N <= WG * NPT;
    for (int base = 0; base < N; base += WG * NPT) {
#pragma unroll
        for (int t = 0; t < NPT; ++t) {
            const int k = base + (int)threadIdx.x + WG * t;
            if (k >= N) continue;
            double a = one_chunk ? ak[t] : ld_sc1(Acol + k);
            if (sym) a = 0.5 * (a + ld_sc1(Acol + ((N - k) & (N - 1))));          // real-pair transforms: |Z|^2 symmetrised
            const int ks = k <= lb.K ? k : (k >= N - lb.K ? k - N : 99);
            double Br = 0.0, Bi = 0.0, wr = 0.0, wi = 0.0;
            if (ks != 99) {                                                      // (2 K + 1 bins of the whole block)
                Br = tot_sh[2 + 2 * (ks + 3)];
                Bi = tot_sh[3 + 2 * (ks + 3)];
#pragma unroll
                for (int u = 0; u < 7; ++u)
                    if (u == ks + 3) {
                        wr = lb.wr[u];
                        wi = lb.wi[u];
                    }
            }
            if constexpr (EXPORT) {
                out[k] = a;
                out[N + 2 * k] = Br;
                out[N + 2 * k + 1] = Bi;
                out[3 * N + 2 * k] = mr * Br + mi * Bi;          // conj(mu0) B
                out[3 * N + 2 * k + 1] = mr * Bi - mi * Br;
            } else {
                const int slot = bin_slot(k, N, sided);
                if (slot >= 0) {
                    const double er = dr * wr - di * wi, ei = dr * wi + di * wr;       // d W[k]
                    const double p = a - 2.0 * (er * Br + ei * Bi) + (double)M * (er * er + ei * ei);
                    out[slot] = p * scale * (bin_doubled(k, N, sided) ? 2.0 : 1.0);
                }
            }
        }
    }
#if SP_OPF_VARIANT & 4
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        OPF_STAMP(ts_end);
        const unsigned nl = g_opf_launches++;
        if (nl % 50 == 49) {
            unsigned long long tmin = ~0ull, tmax0 = 0, tmax1 = 0;
            for (unsigned b = 0; b < gridDim.x; ++b) {
                const unsigned long long a0 = __hip_atomic_load(&g_opf_t0[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long a1 = __hip_atomic_load(&g_opf_t1[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                tmin = a0 < tmin ? a0 : tmin;
                tmax0 = a0 > tmax0 ? a0 : tmax0;
                tmax1 = a1 > tmax1 ? a1 : tmax1;
            }
            // 100 MHz ticks -> ns x 10
            printf("opf stamps (x10 ns from the first block's start): last block start %llu | all phase-1 done %llu | ticket known %llu | "
                   "loads landed %llu | math %llu | reduced %llu | end %llu\n", tmax0 - tmin, tmax1 - tmin, ts_tick - tmin,
                   ts_loaded - tmin, ts_math - tmin, ts_red - tmin, ts_end - tmin);
        }
    }
#endif
    if (threadIdx.x == 0) {
        if constexpr (EXPORT) {
            double *sc = out + 5 * (int64_t)N;
            sc[0] = (double)M * mr;
            sc[1] = (double)M * mi;
            sc[2] = (double)M * (mr * mr + mi * mi);
            sc[3] = tot_r + (double)nmean * mr;
            sc[4] = tot_i + (double)nmean * mi;
            sc[5] = (double)M;
            sc[6] = (double)nmean;
            sc[7] = 0.0;
        }
        // ready for the next launch (plain stores: the atomic form cost 1.6 us at the kernel boundary)
    }
    if (threadIdx.x < 9) ticket[16 * threadIdx.x] = 0u;
    if constexpr (EXPORT) {
        if (prev.st) {          // k_op_apply's arithmetic on the previous step's summed state (same N)
            const double *st = prev.st;
            const double *sc = st + 5 * (int64_t)N;
            const double Mt = sc[5], nt = sc[6];
            const double gr = sc[3] / nt, gi = sc[4] / nt;
            const double sq = (gr * gr + gi * gi) * Mt - 2.0 * (gr * sc[0] + gi * sc[1]) + sc[2];
            for (int k = threadIdx.x; k < N; k += WG) {
                const int slot = bin_slot(k, N, prev.sided);
                if (slot < 0) continue;
                const double b_r = st[N + 2 * k], b_i = st[N + 2 * k + 1], cr = st[3 * N + 2 * k], ci = st[3 * N + 2 * k + 1];
                const double d_r = gr * b_r + gi * b_i - cr, d_i = gr * b_i - gi * b_r - ci;            // conj(mu) B - C
                const double wr = prev.Wf[k].x, wi = prev.Wf[k].y;
                const double p = st[k] - 2.0 * (wr * d_r + wi * d_i) + (wr * wr + wi * wi) * sq;
                prev.out[slot] = p * prev.scale * (bin_doubled(k, N, prev.sided) ? 2.0 : 1.0);
            }
        }
    }
}

// Real input STFT, two frames per transform: z = f_g + i f_{g+1};  X_g = (Z[k] + conj Z[n-k]) / 2,
// X_{g+1} = (Z[k] - conj Z[n-k]) / (2i).  The mirror comes from one more LDS exchange (linear image, reversed read).
// Power-of-two n only (mirror index by masking); the other lengths use k_stft.
// blockIdx.y = channel: x += y*x_cs samples, out += y*out_cs elements, trend += 4*y (pseg only for one channel)
// SHIFT > 0 (hop = SHIFT * T, a whole number of register slots): the samples are carried in registers from pair to pair --
// frame b is frame a shifted by SHIFT slots and the next pair starts 2 SHIFT slots on, so only 2 SHIFT new slots per thread
// and pair are loaded (8 instead of 32 at 75 % overlap), one pair ahead of their use; every sample is loaded once.
#ifndef SP_STFT_MINWAVES
#define SP_STFT_MINWAVES 1
#endif
#ifndef SP_STFT_EU
#define SP_STFT_EU SP_STFT_MINWAVES
#endif
// (the FAST form at 1024 / 2048 points takes 174-183 VGPRs with the window in registers; held to 168 for three workgroups per CU,
//  SP_STFT_FAST_EU=3, it spills 6-13 registers and measured 0.40-0.46 ms against 0.32-0.34 at two per CU for cfg3, tools/stft_ab.sh.
//  With the window in LDS (SP_STFT_WLDS, the default) it takes 155-163 and runs three per CU unforced: 0.282 -> 0.275 ms sustained,
//  0.318 -> 0.296 isolated, tools/stftw_ab.sh)
#ifndef SP_STFT_FAST_EU
#define SP_STFT_FAST_EU SP_STFT_EU
#endif
#ifndef SP_STFT_WLDS
#define SP_STFT_WLDS 1          // FAST form: window in LDS (158 VGPRs, three workgroups per CU; 0: in registers, 178, two)
#endif
// FAST = 1: the one-sided complex spectrogram without the per-frame time-domain power (sided == SIDED_ONE, out_power == 0, pseg ==
// null: spectrogram.stft's shape, cfg3) as compile-time facts -- only the slots t < R/2 hold wanted bins (k < N/2), so half of the
// mirror reads, of the X_a / X_b arithmetic and of the store addresses disappear, and so do the power sums.
template <int N, bool LIN, int SHIFT = 0, int FAST = 0>
__global__ __launch_bounds__(WgCfg<N>::WG) __attribute__((amdgpu_waves_per_eu((FAST && WgCfg<N>::WG == 256 && N <= 2048) ? SP_STFT_FAST_EU : SP_STFT_EU, 8))) void k_stft_rp(const float *__restrict__ x, const float *__restrict__ win,
                                                           int hop, int64_t nframes, int64_t ppg,
                                                           const float *__restrict__ trend, XfTables tb, int sided,
                                                           float amp, int out_power, void *__restrict__ out,
                                                           double *__restrict__ pseg, int64_t x_cs, int64_t out_cs,
                                                           int out_ld /* row pitch of `out` in elements; 0: nbins */) {
    using X = XfPow2<N>;
    SP_KERNEL_PROLOGUE(X)
    // SP_STFT_WLDS (FAST form): the window lives in LDS behind the transform images, [q][thread] float4 = slots 4q .. 4q + 3 of a
    // thread (conflict-free 16-byte reads), instead of 16 registers -- what the form lacks to fit three workgroups per CU
    constexpr bool WLDS = FAST && SP_STFT_WLDS;
    float w[WLDS ? 1 : C::R];
    float4 *w4 = reinterpret_cast<float4 *>(smem + C::FPW * C::LDS_PER);
    if constexpr (WLDS) {
        if (grp == 0) {
#pragma unroll
            for (int q = 0; q < C::R / 4; ++q)
                w4[q * C::T + tid] = make_float4(win[tid + C::T * (4 * q)], win[tid + C::T * (4 * q + 1)], win[tid + C::T * (4 * q + 2)],
                                                 win[tid + C::T * (4 * q + 3)]);
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int t = 0; t < C::R; ++t) w[t] = win[tid + C::T * t];
    }
    x += (int64_t)blockIdx.y * x_cs;
    // out_power 0: complex rows [frame][bin]; 1: float power; 2: complex, the two frames of a pair side by side and the
    // (<= 64) channels interleaved per group of 8 bins, [pair][bin group][channel][8][2] (one 16-byte store per bin; the layout
    // the bf16 contraction of cfg5 reads, k_csdm_bf16.hip; the channel is blockIdx.y, out_cs unused, out_ld a multiple of 8)
    if (out_power == 1) out = reinterpret_cast<float *>(out) + (int64_t)blockIdx.y * out_cs;
    else if (out_power != 2) out = reinterpret_cast<cf *>(out) + (int64_t)blockIdx.y * out_cs;
    const Trend tr = load_trend(trend + 4 * blockIdx.y);
    const int nb = out_ld > 0 ? out_ld : nbins_of(n, sided);
    const int64_t npairs = (nframes + 1) / 2;
    const int64_t gid = (int64_t)blockIdx.x * C::FPW + grp;
    const int64_t p0 = gid * ppg;
    // software pipeline: the samples of the next frame pair are loaded while this pair is transformed
    auto fetch = [&](int64_t p, cf (&dst)[C::R]) {
        const int64_t ga = 2 * (p < npairs ? p : npairs - 1);
        const int64_t base_a = ga * hop, base_b = (ga + 1 < nframes ? ga + 1 : ga) * hop;
#pragma unroll
        for (int t = 0; t < C::R; ++t) dst[t] = mk(x[base_a + tid + C::T * t], x[base_b + tid + C::T * t]);
    };
    constexpr int RS = C::R + SHIFT;                      // slots held: frame a = 0..R-1, frame b = SHIFT..R+SHIFT-1
    const int64_t span = (nframes - 1) * (int64_t)hop + n;
    float ring[SHIFT > 0 ? RS : 1], inc[SHIFT > 0 ? 2 * SHIFT : 1];
    // new slots of pair q (those pair q - 1 does not hold), clamped to the signal
    auto fetch_new = [&](int64_t q) __attribute__((always_inline)) {
        const int64_t b0 = 2 * q * (int64_t)hop + (int64_t)C::T * (RS - 2 * SHIFT);      // uniform
        if (b0 + (int64_t)C::T * 2 * SHIFT <= span) {
            const float *xb = x + b0;                            // whole range inside the signal: scalar base + lane offset
#pragma unroll
            for (int j = 0; j < 2 * SHIFT; ++j) inc[j] = xb[tid + C::T * j];
        } else {
#pragma unroll
            for (int j = 0; j < 2 * SHIFT; ++j) {
                const int64_t idx = b0 + tid + (int64_t)C::T * j;
                inc[j] = x[idx < span ? idx : span - 1];
            }
        }
    };
    cf nxt[SHIFT > 0 ? 1 : C::R];
    if constexpr (SHIFT > 0) {
        const int64_t b = 2 * p0 * (int64_t)hop + tid;
#pragma unroll
        for (int s = 0; s < RS; ++s) {
            const int64_t idx = b + (int64_t)C::T * s;
            ring[s] = x[idx < span ? idx : span - 1];
        }
        fetch_new(p0 + 1);
    } else {
        fetch(p0, nxt);
    }
    for (int64_t i = 0; i < ppg; ++i) {
        const int64_t p = p0 + i;
        const bool act = p < npairs;
        const int64_t ga = 2 * (act ? p : npairs - 1);
        const bool has_b = ga + 1 < nframes;
        const int64_t base_a = ga * hop, base_b = (has_b ? ga + 1 : ga) * hop;
        cf v[C::R];
        float pwa = 0.f, pwb = 0.f;
        if constexpr (SHIFT > 0) {
#pragma unroll
            for (int t = 0; t < C::R; ++t) v[t] = mk(ring[t], ring[t + SHIFT]);
            // advance to pair p + 1 (its new slots arrived during the previous pair), then ask for those of pair p + 2
#pragma unroll
            for (int s = 0; s < RS - 2 * SHIFT; ++s) ring[s] = ring[s + 2 * SHIFT];
#pragma unroll
            for (int j = 0; j < 2 * SHIFT; ++j) ring[RS - 2 * SHIFT + j] = inc[j];
            fetch_new(p + 2);
        } else {
#pragma unroll
            for (int t = 0; t < C::R; ++t) v[t] = nxt[t];
            fetch(p + 1, nxt);
        }
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int j = tid + C::T * t;
            const cf a = detrended<LIN>(mk(v[t].x, 0.f), tr, base_a + j);
            const cf b = detrended<LIN>(mk(v[t].y, 0.f), tr, base_b + j);
            float wt;
            if constexpr (WLDS) {
                const float4 wq = w4[(t / 4) * C::T + tid];
                wt = (t % 4) == 0 ? wq.x : ((t % 4) == 1 ? wq.y : ((t % 4) == 2 ? wq.z : wq.w));
            } else {
                wt = w[t];
            }
            v[t] = mk(wt * a.x, wt * b.x);
            if constexpr (!FAST) {
                const float e = (j == 0 || j == N - 1) ? 0.5f : 1.f;
                pwa += e * v[t].x * v[t].x;
                pwb += e * v[t].y * v[t].y;
            }
        }
        if (!FAST && pseg != nullptr) {        // one atomic per wave and frame instead of one per thread
            constexpr int W = C::T < 64 ? C::T : 64;
            pwa = group_lane_sum<W>(pwa);
            pwb = group_lane_sum<W>(pwb);
            if (act && (tid & (W - 1)) == 0) {
                atomicAdd(&pseg[ga], (double)pwa);
                if (has_b) atomicAdd(&pseg[ga + 1], (double)pwb);
            }
        }
        fwd_row(xf, v, lds, tid, N);
        // mirror exchange
        __syncthreads();
#pragma unroll
        for (int t = 0; t < C::R; ++t) lds[tid + C::T * t] = v[t];
        __syncthreads();
        if constexpr (FAST) {
            if (act) {
                cf *oa = reinterpret_cast<cf *>(out) + ga * nb + tid, *ob = oa + nb;
                const float a2 = amp * 1.41421356237309504880f;
#pragma unroll
                for (int t = 0; t < C::R / 2; ++t) {                   // k = tid + T t < N/2: the one-sided bins, slot = k
                    const int k = tid + C::T * t;
                    const cf zm = lds[(N - k) & (N - 1)];
                    const cf z = v[t];
                    const float a = (k >= 1 && k <= N / 2 - 2) ? a2 : amp;             // [1:-1] of the cropped array is doubled
                    st_stream(oa + C::T * t, mk(a * 0.5f * (z.x + zm.x), a * 0.5f * (z.y - zm.y)));
                    if (has_b) st_stream(ob + C::T * t, mk(a * 0.5f * (z.y + zm.y), -a * 0.5f * (z.x - zm.x)));
                }
            }
        } else if (act) {
#pragma unroll
            for (int t = 0; t < C::R; ++t) {
                const int k = tid + C::T * t;
                const int slot = bin_slot(k, N, sided);
                if (slot < 0) continue;
                const cf zm = lds[(N - k) & (N - 1)];
                const cf z = v[t];
                const cf xa = mk(0.5f * (z.x + zm.x), 0.5f * (z.y - zm.y));          // (Z + conj Zm)/2
                const cf xb = mk(0.5f * (z.y + zm.y), -0.5f * (z.x - zm.x));         // (Z - conj Zm)/(2i)
                if (out_power == 2) {
                    const float a = bin_doubled(k, N, sided) ? amp * 1.41421356237309504880f : amp;
                    const float hb = has_b ? a : 0.f;
                    // [pair][bin group of 8][channel slot of 64][bin in group][frame of the pair]: for one (pair, group) the
                    // 64 channels' 128-byte lines are 8 KiB contiguous -- what one tile row of k_csdm_bf16 streams
                    st_stream(reinterpret_cast<float4 *>(out) + ((((ga / 2) * (int64_t)(nb / 8) + slot / 8) * 64 + blockIdx.y) * 8) + (slot & 7),
                              make_float4(a * xa.x, a * xa.y, hb * xb.x, hb * xb.y));
                } else if (out_power) {
                    st_stream(reinterpret_cast<float *>(out) + ga * nb + slot, amp * cnorm(xa));
                    if (has_b) st_stream(reinterpret_cast<float *>(out) + (ga + 1) * nb + slot, amp * cnorm(xb));
                } else {
                    const float a = bin_doubled(k, N, sided) ? amp * 1.41421356237309504880f : amp;
                    st_stream(reinterpret_cast<cf *>(out) + ga * nb + slot, a * xa);
                    if (has_b) st_stream(reinterpret_cast<cf *>(out) + (ga + 1) * nb + slot, a * xb);
                }
            }
        }
    }
}

// tiled transpose [rows][cols] -> [cols][rows] through LDS
template <typename E>
__global__ void k_transpose(const E *__restrict__ in, E *__restrict__ out, int64_t rows, int64_t cols) {
    __shared__ E tile[32][33];
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int64_t r = r0 + j, c = c0 + threadIdx.x;
        if (r < rows && c < cols) tile[j][threadIdx.x] = in[r * cols + c];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int64_t c = c0 + j, r = r0 + threadIdx.x;
        if (r < rows && c < cols) out[c * rows + r] = tile[threadIdx.x][j];
    }
}

// complex transpose with out = scale * (conj ? conj(in) : in): first / last pass of the large FFT
// blockIdx.z: matrix of a batch (contiguous rows*cols apart)
static __global__ void k_transpose_c(const cf *__restrict__ in, cf *__restrict__ out, int64_t rows, int64_t cols,
                                     int conj, float scale) {
    __shared__ cf tile[32][33];
    in += (int64_t)blockIdx.z * rows * cols;
    out += (int64_t)blockIdx.z * rows * cols;
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    const float sg = conj ? -scale : scale;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int64_t r = r0 + j, c = c0 + threadIdx.x;
        if (r < rows && c < cols) {
            const cf a = in[r * cols + c];
            tile[j][threadIdx.x] = mk(scale * a.x, sg * a.y);
        }
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int64_t c = c0 + j, r = r0 + threadIdx.x;
        if (r < rows && c < cols) out[c * rows + r] = tile[threadIdx.x][j];
    }
}

// ---- elementwise pieces of the long (multi-kernel) paths ----------------------------------------
// out[i] = i < n_in ? (x[i] - mean, 0) : 0   for i < L   (real -> zero-padded complex)
static __global__ void k_pack_real(const float *__restrict__ x, int64_t n_in, const double *__restrict__ mean, int64_t L,
                                   cf *__restrict__ out) {
    const float m = mean ? (float)mean[0] : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = i < n_in ? mk(x[i] - m, 0.f) : mk(0.f, 0.f);
}
// out[i] = a[i] * b[i] (optionally conj(a*b)), i < n
// blockIdx.y: row of a batch (a and out n apart, b shared)
static __global__ void k_cmul_vec(const cf *__restrict__ a, const cf *__restrict__ b, int64_t n, int conj_out,
                                  cf *__restrict__ out) {
    a += (int64_t)blockIdx.y * n;
    out += (int64_t)blockIdx.y * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const cf p = cmul(a[i], b[i]);
        out[i] = conj_out ? cconj(p) : p;
    }
}
// Bluestein pre-multiply with zero padding: out[i] = i < n ? in[i]*chirp[i] : 0, i < L  (conj_in: use conj(in))
// blockIdx.y: row of a batch (in n apart, out L apart)
static __global__ void k_blue_pre(const cf *__restrict__ in, const cf *__restrict__ chirp, int64_t n, int64_t L,
                                  int conj_in, cf *__restrict__ out) {
    in += (int64_t)blockIdx.y * n;
    out += (int64_t)blockIdx.y * L;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < n) {
            const cf a = in[i];
            out[i] = cmul(conj_in ? cconj(a) : a, chirp[i]);
        } else {
            out[i] = mk(0.f, 0.f);
        }
    }
}
// Bluestein post-multiply: out[i] = scale * conj?(in[i] * chirp[i]),  i < n
// blockIdx.y: row of a batch (in in_ld apart, out n apart)
static __global__ void k_blue_post(const cf *__restrict__ in, const cf *__restrict__ chirp, int64_t n, int conj_out,
                                   float scale, cf *__restrict__ out, int64_t in_ld) {
    in += (int64_t)blockIdx.y * in_ld;
    out += (int64_t)blockIdx.y * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const cf p = cmul(in[i], chirp[i]);
        out[i] = mk(scale * p.x, conj_out ? -scale * p.y : scale * p.y);
    }
}
// analytic-signal mask (hilbert.py:63-64) in place: k=0 and k=nyq x1, 1..nyq-1 x2, > nyq x0
static __global__ void k_hilbert_mask(cf *__restrict__ X, int64_t n) {
    const int64_t nyq = (n & 1) ? (n + 1) / 2 : n / 2;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const float h = (k == 0 || k == nyq) ? 1.f : (k < nyq ? 2.f : 0.f);
        X[k] = h * X[k];
    }
}
// c[ch][n] = sum_g detrended(x[ch][g*hop + n]), n < nfft: the time-domain sum of all frames of each channel.  By linearity
// sum_g FFT(win * frame_g) = FFT(win * c): the mean spectrum of the nT-model branch of fft_pwelch (fft_analysis.py:346-393)
// without writing one spectrum.  grid (ceil(nfft/256), frame slices, channels); every slice writes its own partial
// part[slice][ch][n][2] (float64), k_frame_sum_reduce adds the slices in a fixed order: deterministic (round 1 used
// float64 atomics).
template <bool LIN>
static __global__ void k_frame_sum(const void *__restrict__ x, int cplx, int64_t x_ld, int nfft, int hop, int64_t nframes,
                                   const float *__restrict__ trend, double *__restrict__ part) {
    const int n = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int ch = blockIdx.z;
    const int64_t per = (nframes + gridDim.y - 1) / gridDim.y;
    const int64_t g0 = (int64_t)blockIdx.y * per, g1 = g0 + per < nframes ? g0 + per : nframes;
    if (n >= nfft) return;
    const Trend tr = load_trend(trend + 4 * ch);
    const int64_t off = (int64_t)ch * x_ld;
    double sr = 0.0, si = 0.0;
    for (int64_t g = g0; g < g1; ++g) {
        const int64_t i = g * hop + n;
        const cf v = detrended<LIN>(load_sample(x, off + i, cplx != 0), tr, i);
        sr += (double)v.x;
        si += (double)v.y;
    }
    double *p = part + 2 * (((int64_t)blockIdx.y * gridDim.z + ch) * nfft + n);
    p[0] = sr;
    p[1] = si;
}
static __global__ void k_frame_sum_reduce(const double *__restrict__ part, int slices, int64_t count /* nch*nfft*2 */,
                                          double *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    double s = 0.0;
    for (int q = 0; q < slices; ++q) s += part[(int64_t)q * count + e];
    out[e] = s;
}
// X[k] *= H[k] in place (long-row form of sp_spectral_filter)
static __global__ void k_spec_mul(cf *__restrict__ X, const cf *__restrict__ H, int64_t n) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
        X[k] = cmul(X[k], H[k]);
}
// z = (x1-m1) + i (x2-m2), zero-padded to L;  mom[0]=m1, mom[1]=m2
static __global__ void k_xc_pack(const float *__restrict__ x1, const float *__restrict__ x2, int64_t n, int64_t L,
                                 const double *__restrict__ mom, cf *__restrict__ z) {
    const float m1 = (float)mom[0], m2 = (float)mom[1];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (int64_t)gridDim.x * blockDim.x)
        z[i] = i < n ? mk(x1[i] - m1, x2[i] - m2) : mk(0.f, 0.f);
}
// R[k] = A conj(B) from Z = FFT(a + i b):  Im(Z[k] Z[L-k])/2 + i (|Z[k]|^2 - |Z[L-k]|^2)/4 ; stored CONJUGATED
static __global__ void k_xc_mid(const cf *__restrict__ Z, int64_t L, cf *__restrict__ R) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < L; k += (int64_t)gridDim.x * blockDim.x) {
        const cf z = Z[k], zm = Z[(L - k) & (L - 1)];
        const cf zz = cmul(z, zm);
        R[k] = mk(0.5f * zz.y, -0.25f * (cnorm(z) - cnorm(zm)));
    }
}
// ccf with a half-length inverse (the correlation is real): from Z = FFT_L(a + i b), R(k) = A conj(B) as in k_xc_mid, and
// Z'[k] = ((R(k) + conj R(M-k)) + i conj(w) (R(k) - conj R(M-k)))/2, M = L/2, w = exp(-2 pi i k / L): the M-point spectrum of
// z'[n] = r[2n] + i r[2n+1].  Stored CONJUGATED (the inverse runs as a forward transform of the conjugate).
static __global__ void k_xc_mid_half(const cf *__restrict__ Z, int64_t L, BigTw bt, cf *__restrict__ Zp) {
    const int64_t M = L / 2;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < M; k += (int64_t)gridDim.x * blockDim.x) {
        const cf a = Z[k], am = Z[(L - k) & (L - 1)], b = Z[M - k], bm = Z[M + k];
        const cf za = cmul(a, am), zb = cmul(b, bm);
        const cf rk = mk(0.5f * za.y, 0.25f * (cnorm(a) - cnorm(am)));
        const cf rmc = mk(0.5f * zb.y, -0.25f * (cnorm(b) - cnorm(bm)));              // conj R(M-k)
        const cf w = cmul(bt.hi[k >> bt.lb], bt.lo[k & ((1 << bt.lb) - 1)]);          // W_L^k
        const cf s = rk + rmc, d = rk - rmc;
        const cf t = cmul(cconj(w), d);                                               // i t = (-t.y, t.x)
        Zp[k] = mk(0.5f * (s.x - t.y), -0.5f * (s.y + t.x));
    }
}
// half-length Hilbert, middle step, in place: Z = FFT_M(x[2n] + i x[2n+1]) (M = N/2) -> Z'[k] = (conj(w)(Z[k] + conj Z[M-k]) -
// w (Z[k] - conj Z[M-k]))/2, w = exp(-2 pi i k / N), Z'[0] = 0: the half-length spectrum of y = Im(analytic signal), i.e. the
// real-FFT split, the analytic mask (hilbert.py:63-64: DC and Nyquist contribute to the real part only) and the inverse
// real-FFT merge in one step.  A thread owns the pair (k, M-k).
static __global__ void k_hilbert_mid(cf *__restrict__ Z, int64_t M, BigTw bt) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= M / 2; k += (int64_t)gridDim.x * blockDim.x) {
        if (k == 0) {
            Z[0] = mk(0.f, 0.f);
            continue;
        }
        const int64_t km = M - k;
        const cf a = Z[k], b = Z[km];
        const cf w = cmul(bt.hi[k >> bt.lb], bt.lo[k & ((1 << bt.lb) - 1)]);          // W_N^k
        const cf p = mk(a.x + b.x, a.y - b.y), q = mk(a.x - b.x, a.y + b.y);          // a + conj b, a - conj b
        const cf r = cmul(cconj(w), p) - cmul(w, q);
        Z[k] = mk(0.5f * r.x, 0.5f * r.y);
        if (km != k) {
            // for M-k: w' = -conj(w), p' = conj p, q' = -conj q  ->  conj(w') p' - w' q' = -w conj(p) - conj(w) conj(q)
            const cf r2 = cmul(w, cconj(p)) + cmul(cconj(w), cconj(q));
            Z[km] = mk(-0.5f * r2.x, -0.5f * r2.y);
        }
    }
}
// co[j], j < 2n-1, 'full' order from r = real(FFT(conj R))/L:  lag >= 0 -> r[lag], lag < 0 -> r[L+lag]; times mom[2]
static __global__ void k_xc_out(const cf *__restrict__ r, int64_t n, int64_t L, const double *__restrict__ mom,
                                float *__restrict__ co) {
    const float nrm = (float)(mom[2] / (double)L);
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < 2 * n - 1; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t lag = j - (n - 1);
        co[j] = nrm * r[lag >= 0 ? lag : L + lag].x;
    }
}

// ------------------------------------------------------------------------------------------
// cfg5  full cross-spectral-density matrix  G[k][i][j] = sum_g X_i[g,k] conj(X_j[g,k])
// (generalises the reference x channel loop fft_analysis.py:387-393 / HeatPulse_Funcs.py:576-583).
// Stage 1: k_stft per channel -> Xs[c][g][k].  Stage 2: k_csdm_transpose -> Xt[k][c][g] so that, for one bin, every
// channel's frames are contiguous.  Stage 3: k_csdm_gemm, one workgroup per (bin, 64x64 channel block): frames are
// staged through LDS 32 at a time, each thread owns a 4x4 tile of the block in registers (16 complex accumulators,
// 64 FMA per 8 LDS loads -> VALU-bound), and adds its float sums into the float64 matrix in HBM once per chunk.
// ------------------------------------------------------------------------------------------
// Xs[c][g][k] (k fastest, nb per frame) -> Xt[k][c][g]  for g < mc
static __global__ void k_csdm_transpose(const cf *__restrict__ Xs, cf *__restrict__ Xt, int nch, int64_t mc, int nb) {
    __shared__ cf tile[32][33];
    const int c = blockIdx.z;
    const int64_t k0 = (int64_t)blockIdx.x * 32, g0 = (int64_t)blockIdx.y * 32;
    const cf *src = Xs + (int64_t)c * mc * nb;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int64_t gg = g0 + j, k = k0 + threadIdx.x;
        if (gg < mc && k < nb) tile[j][threadIdx.x] = src[gg * nb + k];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int64_t k = k0 + j, gg = g0 + threadIdx.x;
        if (gg < mc && k < nb) Xt[(k * nch + c) * mc + gg] = tile[threadIdx.x][j];
    }
}

#define SP_CM_B 64      // channel block
#define SP_CM_F 32      // frames staged per step
#define SP_CM_P 66      // LDS pitch (complex) of one staged frame: 16-byte aligned rows, reads of 4 consecutive
                        // channels per lane are two conflict-free ds_read_b128, staging writes are 2-way at worst
// blockIdx.z = frame slice [z*fs, (z+1)*fs) of the chunk; with more than one slice the float64 adds are atomic.
// Staged image: A[f][channel] (channel fastest).  Global loads for step s+1 are issued before the FMAs of step s.
static __global__ __launch_bounds__(256) void k_csdm_gemm(const cf *__restrict__ Xt, int nch, int64_t mc, int nblk,
                                                           double *__restrict__ G /*[nb][nch][nch][2]*/, int64_t fs) {
    __shared__ __attribute__((aligned(16))) cf Ai[SP_CM_F][SP_CM_P], Aj[SP_CM_F][SP_CM_P];
    const int k = blockIdx.x;
    const int bi = blockIdx.y / nblk, bj = blockIdx.y % nblk;
    if (bj < bi) return;                                   // Hermitian: the mirror block is filled at the end
    const int ti = threadIdx.x / 16, tj = threadIdx.x % 16;
    cf acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = mk(0.f, 0.f);
    const cf *base = Xt + (int64_t)k * nch * mc;
    const int64_t gbeg = (int64_t)blockIdx.z * fs, gend = gbeg + fs < mc ? gbeg + fs : mc;
    // staging assignment: element e = threadIdx.x + 256*q  ->  (row = e / 32, f = e % 32): a wave reads 2 rows x 32
    // consecutive frames (256 B each) from HBM
    constexpr int NQ = SP_CM_B * SP_CM_F / 256;            // 8
    cf ri[NQ], rj[NQ];
    auto fetch = [&](int64_t g0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = threadIdx.x + 256 * q;
            const int row = e / SP_CM_F, f = e % SP_CM_F;
            const int64_t gg = g0 + f;
            const int ci = bi * SP_CM_B + row, cj = bj * SP_CM_B + row;
            const bool oki = ci < nch && gg < gend, okj = cj < nch && gg < gend;
            const cf a = base[(int64_t)(oki ? ci : 0) * mc + (oki ? gg : gbeg)];
            const cf b = base[(int64_t)(okj ? cj : 0) * mc + (okj ? gg : gbeg)];
            ri[q] = oki ? a : mk(0.f, 0.f);
            rj[q] = okj ? b : mk(0.f, 0.f);
        }
    };
    if (gbeg < gend) fetch(gbeg);
    for (int64_t g0 = gbeg; g0 < gend; g0 += SP_CM_F) {
        __syncthreads();                                   // previous step's readers are done
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = threadIdx.x + 256 * q;
            Ai[e % SP_CM_F][e / SP_CM_F] = ri[q];
            Aj[e % SP_CM_F][e / SP_CM_F] = rj[q];
        }
        __syncthreads();
        if (g0 + SP_CM_F < gend) fetch(g0 + SP_CM_F);      // in flight during the FMAs below
#pragma unroll 4
        for (int f = 0; f < SP_CM_F; ++f) {
            cf a[4], b[4];
            const float4 a01 = *reinterpret_cast<const float4 *>(&Ai[f][4 * ti]);
            const float4 a23 = *reinterpret_cast<const float4 *>(&Ai[f][4 * ti + 2]);
            const float4 b01 = *reinterpret_cast<const float4 *>(&Aj[f][4 * tj]);
            const float4 b23 = *reinterpret_cast<const float4 *>(&Aj[f][4 * tj + 2]);
            a[0] = mk(a01.x, a01.y); a[1] = mk(a01.z, a01.w); a[2] = mk(a23.x, a23.y); a[3] = mk(a23.z, a23.w);
            b[0] = mk(b01.x, b01.y); b[1] = mk(b01.z, b01.w); b[2] = mk(b23.x, b23.y); b[3] = mk(b23.z, b23.w);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    // a conj(b) as four chained FMAs (the compiler may not re-associate `acc += p + q`)
                    acc[u][v].x = fmaf(a[u].x, b[v].x, acc[u][v].x);
                    acc[u][v].x = fmaf(a[u].y, b[v].y, acc[u][v].x);
                    acc[u][v].y = fmaf(a[u].y, b[v].x, acc[u][v].y);
                    acc[u][v].y = fmaf(-a[u].x, b[v].y, acc[u][v].y);
                }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = bi * SP_CM_B + 4 * ti + u, j = bj * SP_CM_B + 4 * tj + v;
            if (i < nch && j < nch) {
                double *p = G + (((int64_t)k * nch + i) * nch + j) * 2;
                if (gridDim.z > 1) {
                    atomicAdd(p, (double)acc[u][v].x);
                    atomicAdd(p + 1, (double)acc[u][v].y);
                } else {
                    p[0] += (double)acc[u][v].x;
                    p[1] += (double)acc[u][v].y;
                }
            }
        }
}

// ---- the same contraction on the matrix cores (default path) -------------------------------------------------
// G[k] = X_k X_k^H is GEMM-shaped (64 x M times M x 64 per bin), so it runs on MFMA: v_mfma_f32_32x32x2_f32 keeps
// float32 products and accumulation (the parity tolerance rules out bf16/fp16 operands; a bf16x3 split would be the
// next step).  Layout: Xt2[k][g][c] -- for one bin and frame the channels are contiguous (padded to a multiple of 64
// with zeros, frames padded to a multiple of 32 with zero frames), so a wave fetches its MFMA operands straight from
// HBM with one coalesced global_load_dwordx2 per 32 channels x 2 frames: lane l holds X[c = l%32][frame f + l/32],
// which is exactly the A (32 x 2) and the B (2 x 32) operand layout of the instruction.  No LDS in the loop.
//   Re G_ij = sum re_i re_j + im_i im_j        Im G_ij = sum im_i re_j - re_i im_j
// One workgroup = one bin x one 64-channel superblock pair (blockIdx.y) x one frame slice; its 4 waves (one per SIMD) split the
// frame pairs of the slice, keep 2 accumulators (Re, Im) per 32 x 32 block, prefetch 4 steps ahead, and are summed
// through LDS at the end.  DIAG (superblock with itself): blocks (0,0), (0,1), (1,1) only; the rest is mirrored.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define SP_CMM_PF 4
template <bool DIAG>
static __global__ __launch_bounds__(256) void k_csdm_mfma(const cf *Xt, int nch, int nchp, int64_t mp, int nsb,
                                                           double *__restrict__ G /*[nb][nch][nch][2]*/, int64_t fs,
                                                           int64_t unit0, int slices, int atomic) {
    __shared__ float red[3][16][64];
    constexpr int NBLK = DIAG ? 3 : 4;
    // work unit = (bin, frame slice); blockIdx.x + unit0 enumerates them slice-fastest
    const int64_t unit = unit0 + blockIdx.x;
    const int k = (int)(unit / slices), zslice = (int)(unit % slices);
    int si, sj;
    if (DIAG) {
        si = sj = blockIdx.y;
    } else {
        // blockIdx.y enumerates the pairs si < sj
        int rem = blockIdx.y;
        si = 0;
        while (rem >= nsb - 1 - si) {
            rem -= nsb - 1 - si;
            ++si;
        }
        sj = si + 1 + rem;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31;
    const int64_t gbeg = (int64_t)zslice * fs, gend = gbeg + fs < mp ? gbeg + fs : mp;          // multiples of 32
    if (gbeg >= mp) return;                                                                      // empty slice (uniform)
    const int nsteps = (int)((gend - gbeg) / 8);                                                 // multiple of SP_CMM_PF, >= 4
    // step s of this wave: frames gbeg + 8 s + 2 wave + {0, 1}
    const cf *pa = Xt + ((int64_t)k * mp + gbeg + 2 * wave + half) * nchp + si * 64 + col;
    const cf *pb = Xt + ((int64_t)k * mp + gbeg + 2 * wave + half) * nchp + sj * 64 + col;
    const int64_t step_stride = (int64_t)8 * nchp;
    f32x16 accR[NBLK], accI[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            accR[b][v] = 0.f;
            accI[b][v] = 0.f;
        }
    // operand ring, SP_CMM_PF steps deep.  The loads are inline asm with hand-placed s_waitcnt: with ordinary loads
    // hipcc turns the loop-carried operands back into load-then-use inside one iteration (the IR carries the addresses,
    // not the data), which exposes the whole memory latency at every step.  The body is branch-free: frame padding
    // makes nsteps a multiple of the depth, loads past the end are clamped to the last step and never used.
    v2f a0[SP_CMM_PF], a1[SP_CMM_PF], b0[SP_CMM_PF], b1[SP_CMM_PF];
    const int last = nsteps - 1;
#define SP_GLOAD2(dst, ptr) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
#pragma unroll
    for (int u = 0; u < SP_CMM_PF; ++u) {
        const int64_t o = (int64_t)(u < last ? u : last) * step_stride;
        SP_GLOAD2(a0[u], pa + o);
        SP_GLOAD2(a1[u], pa + o + 32);
        if (!DIAG) {
            SP_GLOAD2(b0[u], pb + o);
            SP_GLOAD2(b1[u], pb + o + 32);
        }
    }
    for (int s0 = 0; s0 < nsteps; s0 += SP_CMM_PF) {
#pragma unroll
        for (int u = 0; u < SP_CMM_PF; ++u) {
            // the oldest slot's loads are complete when only the (SP_CMM_PF - 1) younger slots' loads are outstanding
            if (DIAG) asm volatile("s_waitcnt vmcnt(6)" : "+v"(a0[u]), "+v"(a1[u])::"memory");
            else asm volatile("s_waitcnt vmcnt(12)" : "+v"(a0[u]), "+v"(a1[u]), "+v"(b0[u]), "+v"(b1[u])::"memory");
            const v2f x0 = a0[u], x1 = a1[u];
            const v2f y0 = DIAG ? x0 : b0[u], y1 = DIAG ? x1 : b1[u];
            const float n0 = -x0.x, n1 = -x1.x;
            // blocks 0: (I=0,J=0)  1: (0,1)  2: (1,1)  3: (1,0); consecutive MFMAs use different accumulators
            accR[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.x, y0.x, accR[0], 0, 0, 0);
            accR[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.x, y1.x, accR[1], 0, 0, 0);
            accR[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.x, y1.x, accR[2], 0, 0, 0);
            if constexpr (!DIAG) accR[NBLK - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.x, y0.x, accR[NBLK - 1], 0, 0, 0);
            accI[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.y, y0.x, accI[0], 0, 0, 0);
            accI[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.y, y1.x, accI[1], 0, 0, 0);
            accI[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.y, y1.x, accI[2], 0, 0, 0);
            if constexpr (!DIAG) accI[NBLK - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.y, y0.x, accI[NBLK - 1], 0, 0, 0);
            accR[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.y, y0.y, accR[0], 0, 0, 0);
            accR[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.y, y1.y, accR[1], 0, 0, 0);
            accR[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.y, y1.y, accR[2], 0, 0, 0);
            if constexpr (!DIAG) accR[NBLK - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.y, y0.y, accR[NBLK - 1], 0, 0, 0);
            accI[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(n0, y0.y, accI[0], 0, 0, 0);
            accI[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(n0, y1.y, accI[1], 0, 0, 0);
            accI[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(n1, y1.y, accI[2], 0, 0, 0);
            if constexpr (!DIAG) accI[NBLK - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(n1, y0.y, accI[NBLK - 1], 0, 0, 0);
            // refill the slot for step s0 + u + SP_CMM_PF
            const int sn = s0 + u + SP_CMM_PF;
            const int64_t o = (int64_t)(sn < last ? sn : last) * step_stride;
            SP_GLOAD2(a0[u], pa + o);
            SP_GLOAD2(a1[u], pa + o + 32);
            if (!DIAG) {
                SP_GLOAD2(b0[u], pb + o);
                SP_GLOAD2(b1[u], pb + o + 32);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef SP_GLOAD2
    // sum the four waves through LDS (one 32 x 32 accumulator at a time), then wave 0 adds into the float64 matrix.
    // Accumulator layout of the instruction: register v of lane l is D[i = 8 (v/4) + 4 (l/32) + v%4][j = l%32].
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
        const int bi = (b == 0 || b == 1) ? 0 : 1, bj = (b == 0 || b == 3) ? 0 : 1;
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            f32x16 &acc = part ? accI[b] : accR[b];
            __syncthreads();
            if (wave > 0) {
#pragma unroll
                for (int v = 0; v < 16; ++v) red[wave - 1][v][lane] = acc[v];
            }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float t = (acc[v] + red[0][v][lane]) + (red[1][v][lane] + red[2][v][lane]);
                    const int i = si * 64 + 32 * bi + 8 * (v / 4) + 4 * half + (v % 4), j = sj * 64 + 32 * bj + col;
                    if (i < nch && j < nch) {
                        double *p = G + (((int64_t)k * nch + i) * nch + j) * 2 + part;
                        if (atomic) atomicAdd(p, (double)t);
                        else *p += (double)t;
                    }
                }
            }
        }
    }
}

// ---- fused form: no transposed copy of the spectra -------------------------------------------------------------
// One workgroup of 16 waves owns 16 consecutive bins (wave w <-> bin k0 + w) and reads the STFT output Xs[c][g][k] as
// it lies: a (channel, frame) row of its 16 bins is one 128-byte line (8-bin groups, 64-byte rows, ran at half the
// speed: PMC showed 1.4-2.9x the algorithmic HBM bytes and the MFMA pipe 49 % busy).  Tiles of 4 frames x 64 channels
// x 16 bins (32 KiB) go HBM -> registers -> LDS (double buffered, one barrier per
// tile); every wave then reads ITS bin's operands from LDS in the MFMA layout (lane l: channel l%32, frame parity l/32;
// pitch 17 complex per (frame, channel) row keeps the reads conflict-free) and runs the same 12 (16) MFMAs per frame
// pair as k_csdm_mfma.  Each wave owns its bin's accumulators: no cross-wave reduction.  Work unit = (bin group, frame
// slice), one 1024-thread workgroup per CU (96 accumulator + 32 other registers per lane).  Only for nch <= 64: a pair
// of different 64-channel superblocks needs 128 accumulators.  Bins that do not fill a group of 16 (the Nyquist bin of every
// power-of-two nfft) take the k_csdm_mfma path on a small gathered copy.
#define SP_CMF_BINS 16
#define SP_CMF_F 4        // frames per tile
#define SP_CMF_P 17       // LDS pitch (complex) of one (frame, channel) row
#define SP_CMF_TILE (SP_CMF_F * 64 * SP_CMF_P)      // complex elements per buffer
static __global__ __launch_bounds__(1024) void k_csdm_fused(const cf *Xs, int nch, int64_t m, int nb /* row pitch of Xs */,
                                                             double *__restrict__ G, int64_t fs, int slices, int atomic) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *lds = reinterpret_cast<cf *>(smem_raw);
    constexpr int F = SP_CMF_F;
    const int unit = blockIdx.x;
    const int k0 = (unit / slices) * SP_CMF_BINS, zslice = unit % slices;
    const int64_t gbeg = (int64_t)zslice * fs, gend = gbeg + fs < m ? gbeg + fs : m;
    if (gbeg >= gend) return;
    const int ntiles = (int)((gend - gbeg + F - 1) / F);
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, half = lane >> 5, col = lane & 31;
    // staging: 2 passes x (128 rows x 8 parts of 2 bins); row = frame * 64 + channel.  The channel of a thread is
    // fixed, so is its row base; only 32 arch VGPRs are left beside the 96 accumulators (4 waves per SIMD).
    const int part = t & 7, rowq = t >> 3, cl = rowq & 63, f0 = rowq >> 6;
    const float keepc = cl < nch ? 1.f : 0.f;
    const cf *rowbase = Xs + (int64_t)(cl < nch ? cl : 0) * m * nb + k0 + 2 * part;
    cf *ldst = lds + (f0 * 64 + cl) * SP_CMF_P + 2 * part;
    cf st[2][2];
    float keep[2];
    auto gfetch = [&](int64_t g0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int64_t g = g0 + f0 + 2 * q;
            const bool ok = g < gend;
            const cf *src = rowbase + (ok ? g : gbeg) * nb;          // clamped address; the value is masked at lstore,
            st[q][0] = src[0];                                       // so nothing here waits for the loads
            st[q][1] = src[1];
            keep[q] = ok ? keepc : 0.f;
        }
    };
    auto lstore = [&](int bufsel) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            cf *dst = ldst + bufsel * SP_CMF_TILE + (2 * q * 64) * SP_CMF_P;
            dst[0] = keep[q] * st[q][0];
            dst[1] = keep[q] * st[q][1];
        }
    };
    f32x16 accR[3], accI[3];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            accR[b][v] = 0.f;
            accI[b][v] = 0.f;
        }
    gfetch(gbeg);
    lstore(0);
    __syncthreads();
    const cf *pa0 = lds + (half * 64 + col) * SP_CMF_P + wave;
    for (int it = 0; it < ntiles; ++it) {
        const bool more = it + 1 < ntiles;                    // workgroup-uniform
        if (more) gfetch(gbeg + (int64_t)(it + 1) * F);
        const cf *pa = pa0 + (it & 1) * SP_CMF_TILE;
#pragma unroll
        for (int p = 0; p < F / 2; ++p) {
            const cf x0 = pa[(2 * p * 64) * SP_CMF_P], x1 = pa[(2 * p * 64 + 32) * SP_CMF_P];
            const float n0 = -x0.x;
            // blocks 0: (0,0)  1: (0,1)  2: (1,1); consecutive MFMAs use different accumulators.  On the diagonal blocks
            // Im G = P - P^T with P = Im Re^T: only P is accumulated (10 MFMAs per frame pair instead of 12), the
            // transpose is taken once at the end
            accR[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.x, x0.x, accR[0], 0, 0, 0);
            accR[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.x, x1.x, accR[1], 0, 0, 0);
            accR[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.x, x1.x, accR[2], 0, 0, 0);
            accI[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.y, x0.x, accI[0], 0, 0, 0);
            accI[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.y, x1.x, accI[1], 0, 0, 0);
            accI[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.y, x1.x, accI[2], 0, 0, 0);
            accR[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.y, x0.y, accR[0], 0, 0, 0);
            accR[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.y, x1.y, accR[1], 0, 0, 0);
            accR[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.y, x1.y, accR[2], 0, 0, 0);
            accI[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(n0, x1.y, accI[1], 0, 0, 0);
        }
        if (more) lstore((it + 1) & 1);
        __syncthreads();
    }
    // register v of lane l is D[i = 8 (v/4) + 4 (l/32) + v%4][j = l%32]
    const int k = k0 + wave;
    float *tp = reinterpret_cast<float *>(lds) + wave * (32 * 33);       // this wave's 32 x 32 transpose image (pitch 33)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int bi = b == 2 ? 1 : 0, bj = b == 0 ? 0 : 1;
        if (b != 1) {
            // diagonal block: Im = P - P^T
            __syncthreads();
#pragma unroll
            for (int v = 0; v < 16; ++v) tp[(8 * (v / 4) + 4 * half + (v % 4)) * 33 + col] = accI[b][v];
            __syncthreads();
#pragma unroll
            for (int v = 0; v < 16; ++v) accI[b][v] -= tp[col * 33 + 8 * (v / 4) + 4 * half + (v % 4)];
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int i = 32 * bi + 8 * (v / 4) + 4 * half + (v % 4), j = 32 * bj + col;
            if (i < nch && j < nch) {
                double *p = G + (((int64_t)k * nch + i) * nch + j) * 2;
                if (atomic) {
                    atomicAdd(p, (double)accR[b][v]);
                    atomicAdd(p + 1, (double)accI[b][v]);
                } else {
                    p[0] += (double)accR[b][v];
                    p[1] += (double)accI[b][v];
                }
            }
        }
    }
}

// tail bins for the fused path: Xt2[kk][g][c] = Xs[c][g][kfirst + kk], zero padded (kk < ntail <= 16)
static __global__ void k_csdm_gather_bins(const cf *__restrict__ Xs, cf *__restrict__ Xt, int nch, int nchp, int64_t m, int64_t mp,
                                          int nb /* row pitch of Xs */, int kfirst, int ntail) {
    const int64_t g = blockIdx.x;
    for (int e = threadIdx.x; e < ntail * nchp; e += blockDim.x) {
        const int kk = e / nchp, c = e % nchp;
        const bool ok = c < nch && g < m;
        const cf v = Xs[ok ? ((int64_t)c * m + g) * nb + kfirst + kk : 0];
        Xt[((int64_t)kk * mp + g) * nchp + c] = ok ? v : mk(0.f, 0.f);
    }
}

// Xs[c][g][k] (k fastest) -> Xt2[k][g][c] (c fastest, nchp channels, mp frames; the padding is written as zeros)
static __global__ void k_csdm_transpose_kgc(const cf *__restrict__ Xs, cf *__restrict__ Xt, int nch, int nchp, int64_t m,
                                            int64_t mp, int nb) {
    __shared__ cf tile[32][33];
    const int64_t g = blockIdx.z;
    const int k0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int c = c0 + j, k = k0 + threadIdx.x;
        const bool ok = c < nch && g < m && k < nb;
        const cf v = Xs[ok ? ((int64_t)c * m + g) * nb + k : 0];
        tile[j][threadIdx.x] = ok ? v : mk(0.f, 0.f);
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int k = k0 + j, c = c0 + threadIdx.x;
        if (k < nb) Xt[((int64_t)k * mp + g) * nchp + c] = tile[threadIdx.x][j];
    }
}

// scale, and fill the blocks below the block diagonal from their Hermitian mirrors
// (blk = granularity of the computed upper block triangle: 64 for the VALU kernel, 32 for the MFMA kernel)
static __global__ void k_csdm_finish(double *__restrict__ G, int nch, int nb, double scale, int blk) {
    const int64_t total = (int64_t)nb * nch * nch;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(e % nch), i = (int)((e / nch) % nch);
        if (j / blk >= i / blk) {
            G[2 * e] *= scale;
            G[2 * e + 1] *= scale;
        }
    }
}
static __global__ void k_csdm_mirror(double *__restrict__ G, int nch, int nb, int blk) {
    const int64_t total = (int64_t)nb * nch * nch;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(e % nch), i = (int)((e / nch) % nch);
        const int64_t k = e / ((int64_t)nch * nch);
        if (j / blk < i / blk) {
            const int64_t m = ((k * nch + j) * nch + i);
            G[2 * e] = G[2 * m];
            G[2 * e + 1] = -G[2 * m + 1];
        }
    }
}

// ------------------------------------------------------------------------------------------
// A10  analytic signal (hilbert.py:54-67): fft -> zero [nyq+1:], double [1:nyq) -> ifft, in one
// workgroup per row.  nyq = n/2 (even) or (n+1)/2 (odd): for odd n bin `nyq` is left untouched,
// exactly as the reference does (Q6).
// ------------------------------------------------------------------------------------------
#ifndef SP_HILBERT_WAVES
#define SP_HILBERT_WAVES 1
#endif
#ifndef SP_HILBERT_EU
#define SP_HILBERT_EU 3
#endif
// RESP: the spectrum is multiplied by the table H[0:n] instead of the mask (sp_spectral_filter: fft_deriv's
// wavenumber, fft_analysis.py:1526-1546, or any other frequency response).
// (the mask form takes 169 VGPRs unbounded -- one over what three workgroups of 256 threads per CU allow; held to 168)
template <class X, bool RESP>
__global__ __launch_bounds__(X::C::WG)
    __attribute__((amdgpu_waves_per_eu((!RESP && X::C::WG == 256 && X::EXACT) ? SP_HILBERT_EU : SP_HILBERT_WAVES,
                                       8))) void k_hilbert(const float *__restrict__ x, int64_t n_in, int64_t x_ld,
                                                       int64_t batch, XfTables tb, cf *__restrict__ out,
                                                       const cf *__restrict__ H) {
    SP_KERNEL_PROLOGUE(X)
    const int nyq = nyq_of(n);
    const float inv = 1.f / (float)n;
    const int64_t stride = (int64_t)gridDim.x * C::FPW;
    for (int64_t b0 = (int64_t)blockIdx.x * C::FPW; b0 < batch; b0 += stride) {
        const int64_t b = b0 + grp;
        const bool act = b < batch;
        const int64_t bl = act ? b : batch - 1;
        cf v[C::R];
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int j = tid + C::T * t;
            const float a = x[bl * x_ld + (j < n_in ? j : n_in - 1)];     // clamped, unconditional
            v[t] = mk(j < n_in ? a : 0.f, 0.f);
        }
        fwd_row(xf, v, lds, tid, n);
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int k = tid + C::T * t;
            if constexpr (RESP) {
                const bool in = X::EXACT || k < n;
                const cf p = cmul(v[t], H[in ? k : 0]);
                v[t] = in ? mk(p.x, -p.y) : mk(0.f, 0.f);   // response, then conj for the inverse
            } else {
                const float h = (k == 0 || k == nyq) ? 1.f : (k < nyq ? 2.f : 0.f);
                v[t] = mk(h * v[t].x, -h * v[t].y);      // mask, then conj for the inverse
            }
        }
        fwd_row(xf, v, lds, tid, n);
        if (act) {
#pragma unroll
            for (int t = 0; t < C::R; ++t) {
                const int k = tid + C::T * t;
                if (X::EXACT || k < n) st_stream(out + b * n + k, mk(v[t].x * inv, -v[t].y * inv));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// F1  causal FIR by overlap-save; two real blocks ride in one complex transform (h real).
// Block b yields y[b*Lb : (b+1)*Lb) from x[b*Lb-(P-1) : b*Lb+Lb), Lb = N-(P-1).
// Hs[k] = FFT_N(h)[k] / N  (scale of the inverse folded in).
// The kernel handles block pairs [p_begin, p_end).  EDGE=false: every pair in the range lies wholly inside the
// signal -- no bounds predicates at all (they cost ~500 VALU ops per pair in 64-bit compares and selects);
// EDGE=true: per-sample predicates, used for the first pair and the last few only.
// Inputs of the next pair are loaded while the current pair is transformed.
// ------------------------------------------------------------------------------------------
// (launch bound: at least 2 waves per SIMD, i.e. <= 256 VGPRs -- unbounded, hipcc takes 264 and halves occupancy)
// Round 3: the interior form at 2048 / 4096 points runs THREE workgroups per CU without the next pair's loads in flight (157-163
// VGPRs) instead of two with them (208-215): cfg4 0.61-0.64 -> 0.53-0.58 ms on one box (tools/fir_ab.sh).  SP_FIR_EU=2
// SP_FIR_PREFETCH=1 restore the old form.
#ifndef SP_FIR_EU
#define SP_FIR_EU 3
#endif
#ifndef SP_FIR_PREFETCH
#define SP_FIR_PREFETCH 0
#endif
template <int N, bool EDGE>
__global__ __launch_bounds__(WgCfg<N>::WG)
    __attribute__((amdgpu_waves_per_eu((!EDGE && WgCfg<N>::WG == 256 && N >= 2048) ? SP_FIR_EU : 2, 8))) void k_fftfilt(const float *__restrict__ x, int64_t nsamp, int ntaps,
                                                           const cf *__restrict__ Hs, XfTables tb,
                                                           float *__restrict__ y, int64_t p_begin, int64_t p_end) {
    using X = XfPow2<N>;
    SP_KERNEL_PROLOGUE(X)
    (void)n;
    const int P1 = ntaps - 1;
    const int64_t Lb = N - P1;
    const int lb = (int)Lb;
    const int64_t stride = (int64_t)gridDim.x * C::FPW;
    auto fetch = [&](int64_t p, cf (&dst)[C::R]) {
        const bool act = p < p_end;
        const int64_t s0 = 2 * (act ? p : p_end - 1) * Lb - P1;      // clamped pair: loads stay in range
        if constexpr (!EDGE) {
            const float *b0 = x + s0;
#pragma unroll
            for (int t = 0; t < C::R; ++t) dst[t] = mk(ld_stream(b0 + tid + C::T * t), ld_stream(b0 + lb + tid + C::T * t));
        } else {
            const int64_t s1 = s0 + Lb;
#pragma unroll
            for (int t = 0; t < C::R; ++t) {
                const int m = tid + C::T * t;
                const int64_t i0 = s0 + m, i1 = s1 + m;
                const bool in0 = i0 >= 0 && i0 < nsamp, in1 = i1 >= 0 && i1 < nsamp;
                const float a = x[in0 ? i0 : 0], b = x[in1 ? i1 : 0];      // clamped, unconditional
                dst[t] = mk(in0 ? a : 0.f, in1 ? b : 0.f);
            }
        }
    };
    constexpr bool PF = SP_FIR_PREFETCH || EDGE || WgCfg<N>::WG != 256 || N < 2048;
    cf nxt[C::R];
    if constexpr (PF) fetch(p_begin + (int64_t)blockIdx.x * C::FPW + grp, nxt);
    for (int64_t p0 = p_begin + (int64_t)blockIdx.x * C::FPW; p0 < p_end; p0 += stride) {
        const int64_t p = p0 + grp;
        const bool act = p < p_end;
        const int64_t s0 = 2 * p * Lb - P1, s1 = s0 + Lb;
        cf v[C::R];
        if constexpr (PF) {
#pragma unroll
            for (int t = 0; t < C::R; ++t) v[t] = nxt[t];
            fetch(p + stride, nxt);
        } else {
            fetch(p, v);
        }
        // the filter spectrum (32 KiB, L2-resident) is re-read every pair instead of pinning 32 VGPRs; the opaque
        // zero offset stops hipcc from hoisting the loads out of the loop and spilling
        int hoff = 0;
        asm volatile("" : "+v"(hoff));
        cf H[C::R];
#pragma unroll
        for (int t = 0; t < C::R; ++t) H[t] = Hs[hoff + tid + C::T * t];
        fwd_row(xf, v, lds, tid, N);
#pragma unroll
        for (int t = 0; t < C::R; ++t) v[t] = cconj(cmul(v[t], H[t]));
        fwd_row(xf, v, lds, tid, N);
        if (act) {
            if constexpr (!EDGE) {
                float *y0 = y + s0;
#pragma unroll
                for (int t = 0; t < C::R; ++t) {
                    const int m = tid + C::T * t;
                    if (m >= P1) {
                        st_stream(y0 + m, v[t].x);
                        st_stream(y0 + lb + m, -v[t].y);         // conj of the inverse trick
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < C::R; ++t) {
                    const int m = tid + C::T * t;
                    if (m >= P1) {
                        const int64_t o0 = s0 + m, o1 = s1 + m;
                        if (o0 < nsamp) y[o0] = v[t].x;
                        if (o1 < nsamp) y[o1] = -v[t].y;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// A11  cross-covariance at all lags for n <= N/2 (ccf.py:74-76) in one workgroup:
// z = (x1-m1) + i (x2-m2) zero-padded to N;  A conj(B) = Im(Z[k] Z[N-k])/2 + i (|Z[k]|^2-|Z[N-k]|^2)/4
// mom[0..2] = mean1, mean2, 1/(n*std1*std2)  (device)
// ------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(WgCfg<N>::WG) void k_xcorr(const float *__restrict__ x1, const float *__restrict__ x2,
                                                         int64_t nsamp, const double *__restrict__ mom, XfTables tb,
                                                         float *__restrict__ co) {
    using X = XfPow2<N>;
    SP_KERNEL_PROLOGUE(X)
    (void)n;
    // only group 0 carries data; the others transform zeros so every thread meets the same barriers
    const bool act = grp == 0;
    const float m1 = (float)mom[0], m2 = (float)mom[1];
    const float nrm = (float)mom[2];
    cf v[C::R];
#pragma unroll
    for (int t = 0; t < C::R; ++t) {
        const int i = tid + C::T * t;
        const bool in = act && i < nsamp;
        const float a = x1[in ? i : 0], b = x2[in ? i : 0];
        v[t] = in ? mk(a - m1, b - m2) : mk(0.f, 0.f);
    }
    xf.fwd(v, lds, tid, N);
    // mirror exchange: Z[(N-k)%N]
    __syncthreads();
#pragma unroll
    for (int t = 0; t < C::R; ++t) lds[tid + C::T * t] = v[t];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < C::R; ++t) {
        const int k = tid + C::T * t;
        const cf zm = lds[(N - k) & (N - 1)];
        const cf z = v[t];
        const cf zz = cmul(z, zm);
        // R = A conj(B); feed conj(R) to the forward transform to get the inverse
        v[t] = mk(0.5f * zz.y, -0.25f * (cnorm(z) - cnorm(zm)));
    }
    xf.fwd(v, lds, tid, N);
    if (act) {
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int i = tid + C::T * t;          // r[i] = v.x / N   (imag ~ 0)
            const float r = v[t].x * (nrm / N);
            // 'full' order: j = lag + (n-1);  lag >= 0 -> r[lag], lag < 0 -> r[N+lag]
            if (i < nsamp) co[(nsamp - 1) + i] = r;
            else if (i > N - nsamp) co[i - (N - nsamp + 1)] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------
// reductions in double: sums for mean / variance / least-squares line
// partial[block][8] = sum re, sum im, sum |x|^2, sum i*re, sum i*im, 0, 0, 0   (i = sample index)
// ------------------------------------------------------------------------------------------
#define SP_MOM 5
// blockIdx.y = signal number: samples start at x + y*x_cs, partials at partial + y*gridDim.x*8
template <bool CPLX, bool LIN>
__global__ __launch_bounds__(256) void k_moments_partial(const void *__restrict__ x, int64_t n,
                                                          double *__restrict__ partial, int64_t x_cs) {
    double s[SP_MOM] = {0, 0, 0, 0, 0};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t off = (int64_t)blockIdx.y * x_cs;
    partial += (int64_t)blockIdx.y * gridDim.x * 8;
    auto acc = [&](float re, float im, int64_t i) {
        s[0] += re;
        s[1] += im;
        s[2] += (double)re * re + (double)im * im;
        if constexpr (LIN) {
            const double di = (double)i;
            s[3] += di * re;
            s[4] += di * im;
        }
    };
    // 16-byte loads, four in flight per thread (one dword per thread and iteration kept ~8 KB per CU in flight: 3.1 TB/s);
    // the few samples before the first 16-byte boundary and after the last whole vector go through scalar loads
    constexpr int VEC = CPLX ? 2 : 4;
    constexpr int ESZ = CPLX ? 8 : 4;
    const char *base = reinterpret_cast<const char *>(x) + off * ESZ;
    int64_t head = (int64_t)(((16 - (reinterpret_cast<uintptr_t>(base) & 15)) & 15) / ESZ);
    if (head > n) head = n;
    const int64_t nvec = (n - head) / VEC;
    const float4 *xv = reinterpret_cast<const float4 *>(base + head * ESZ);
    auto accv = [&](const float4 &q, int64_t j) {
        const int64_t i = head + j * VEC;
        if constexpr (CPLX) {
            acc(q.x, q.y, i);
            acc(q.z, q.w, i + 1);
        } else {
            acc(q.x, 0.f, i);
            acc(q.y, 0.f, i + 1);
            acc(q.z, 0.f, i + 2);
            acc(q.w, 0.f, i + 3);
        }
    };
    int64_t j = gtid;
    for (; j + 3 * stride < nvec; j += 4 * stride) {
        const float4 q0 = xv[j], q1 = xv[j + stride], q2 = xv[j + 2 * stride], q3 = xv[j + 3 * stride];
        accv(q0, j);
        accv(q1, j + stride);
        accv(q2, j + 2 * stride);
        accv(q3, j + 3 * stride);
    }
    for (; j < nvec; j += stride) accv(xv[j], j);
    if (blockIdx.x == 0) {
        const int64_t tail0 = head + nvec * VEC;
        if ((int64_t)threadIdx.x < head) {
            const cf a = load_sample(x, off + threadIdx.x, CPLX);
            acc(a.x, a.y, threadIdx.x);
        }
        if (tail0 + (int64_t)threadIdx.x < n) {
            const cf a = load_sample(x, off + tail0 + threadIdx.x, CPLX);
            acc(a.x, a.y, tail0 + threadIdx.x);
        }
    }
    __shared__ double sh[SP_MOM][256];
#pragma unroll
    for (int j = 0; j < SP_MOM; ++j) sh[j][threadIdx.x] = s[j];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
#pragma unroll
            for (int j = 0; j < SP_MOM; ++j) sh[j][threadIdx.x] += sh[j][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x < SP_MOM) partial[blockIdx.x * 8 + threadIdx.x] = sh[threadIdx.x][0];
}

// one block of 256.  out_d[0..1] = mean, out_d[2] = sum|x|^2, out_d[3..4] = sum i*x.
// trend_f[4] (optional): mode 1 -> (mean, 0 slope); mode 2 -> least-squares line m + s*i
// blockIdx.x = signal number (partials / outputs strided accordingly)
static __global__ __launch_bounds__(256) void k_moments_finish(const double *__restrict__ partial, int nblocks, int64_t n,
                                                         int mode, double *__restrict__ out_d,
                                                         float *__restrict__ trend_f) {
    __shared__ double sh[SP_MOM][256];
    partial += (int64_t)blockIdx.x * nblocks * 8;
    if (out_d) out_d += (int64_t)blockIdx.x * 8;
    if (trend_f) trend_f += (int64_t)blockIdx.x * 4;
    double s[SP_MOM] = {0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < nblocks; b += 256) {
#pragma unroll
        for (int j = 0; j < SP_MOM; ++j) s[j] += partial[b * 8 + j];
    }
#pragma unroll
    for (int j = 0; j < SP_MOM; ++j) sh[j][threadIdx.x] = s[j];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
#pragma unroll
            for (int j = 0; j < SP_MOM; ++j) sh[j][threadIdx.x] += sh[j][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double N = (double)n;
        const double mr = sh[0][0] / N, mi = sh[1][0] / N;
        if (out_d) {
            out_d[0] = mr;
            out_d[1] = mi;
            out_d[2] = sh[2][0];
            out_d[3] = sh[3][0];
            out_d[4] = sh[4][0];
        }
        if (trend_f) {
            if (mode == 2 && n > 1) {
                // least squares on i = 0..n-1:  slope = (sum i x - ibar sum x) / sum (i-ibar)^2
                const double ibar = 0.5 * (N - 1.0);
                const double sxx = N * (N * N - 1.0) / 12.0;
                const double sr = (sh[3][0] - ibar * sh[0][0]) / sxx, si = (sh[4][0] - ibar * sh[1][0]) / sxx;
                trend_f[0] = (float)(mr - sr * ibar);
                trend_f[1] = (float)(mi - si * ibar);
                trend_f[2] = (float)sr;
                trend_f[3] = (float)si;
            } else {
                trend_f[0] = (float)mr;
                trend_f[1] = (float)mi;
                trend_f[2] = 0.f;
                trend_f[3] = 0.f;
            }
        }
    }
}

// ccf (ccf.py:74-76): both signals' moment records finished by ONE block and the normalisation record [mean1, mean2, 1 / (n std1
// std2), 0] written behind them -- k_moments_finish x2 + k_xcorr_norm in one launch (two kernel boundaries less per call)
static __global__ __launch_bounds__(256) void k_moments_finish_xc(const double *__restrict__ partial, int nblocks, int64_t n,
                                                            double *__restrict__ out_d, double *__restrict__ xc_out) {
    __shared__ double sh[3][256];
    double mean[2], ssq[2];
    for (int sig = 0; sig < 2; ++sig) {
        const double *pp = partial + (int64_t)sig * nblocks * 8;
        double s0 = 0, s2 = 0;
        for (int b = threadIdx.x; b < nblocks; b += 256) {
            s0 += pp[b * 8 + 0];
            s2 += pp[b * 8 + 2];
        }
        sh[0][threadIdx.x] = s0;
        sh[2][threadIdx.x] = s2;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) {
                sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
                sh[2][threadIdx.x] += sh[2][threadIdx.x + o];
            }
            __syncthreads();
        }
        mean[sig] = sh[0][0] / (double)n;
        ssq[sig] = sh[2][0];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        for (int sig = 0; sig < 2; ++sig) {
            out_d[sig * 8 + 0] = mean[sig];
            out_d[sig * 8 + 1] = 0.0;
            out_d[sig * 8 + 2] = ssq[sig];
            out_d[sig * 8 + 3] = 0.0;
            out_d[sig * 8 + 4] = 0.0;
        }
        const double v1 = ssq[0] / (double)n - mean[0] * mean[0], v2 = ssq[1] / (double)n - mean[1] * mean[1];
        xc_out[0] = mean[0];
        xc_out[1] = mean[1];
        xc_out[2] = 1.0 / ((double)n * sqrt(v1 > 0 ? v1 : 0) * sqrt(v2 > 0 ? v2 : 0));
        xc_out[3] = 0;
    }
}

}   // namespace sp
