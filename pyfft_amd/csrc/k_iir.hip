// k_iir.hip -- exact second-order IIR section  y = lfilter(b, a, x)  on the GPU (row F2 of the scope table: the reference
// only DESIGNS notch / peak biquads, notch_filter.py:175-241; nothing in it applies them -- scipy.signal.lfilter is the
// oracle).  A truncated-impulse-response FIR cannot stand in for a narrow notch (pole radius 0.9995 at w0 = 0.01, Q = 30:
// 28 000 taps for 1e-6), so the recurrence itself is parallelised, exactly:
//
//   y[n] = v[n] - a1 y[n-1] - a2 y[n-2],   v[n] = b0 x[n] + b1 x[n-1] + b2 x[n-2]
//
// is linear in its state s = (y[n-1], y[n-2]).  A run of C samples started from state s ends in  M_C s + f,  f being the
// end state of the same run started from rest, and its outputs are  y_rest[j] + alpha[j] s0 + beta[j] s1  with alpha, beta the
// two homogeneous solutions.  So: every thread runs its C-sample chunk from rest (float64), a workgroup combines its 256
// chunk end states with a log-step scan of the affine maps (all maps share M_C, the scan needs only the powers
// M_C^(2^d)), and tiles are chained the same way one level up.  Three launches:
//   k_biquad_tile<false>   per tile: end state from rest                       (reads x: 4 B / sample)
//   k_biquad_chain         incoming state of every tile (one workgroup, scan over tiles)
//   k_biquad_tile<true>    per tile: outputs with the right incoming state     (reads x, writes y: 8 B / sample)
// = 12 B per sample against 8 B algorithmic; all arithmetic in float64, inputs and outputs float32, so the result
// equals scipy.signal.lfilter on the float32 samples up to the rounding of the output.
#include "launch.h"
namespace sp {

#define SP_IIR_C 32                       /* samples per thread */
#define SP_IIR_WG 256
#define SP_IIR_TILE (SP_IIR_C * SP_IIR_WG)

struct M2 {                               // 2x2 matrix, row major
    double a, b, c, d;
};
struct BiquadPlan {
    double b0, b1, b2, a1, a2;
    double alpha[SP_IIR_C], beta[SP_IIR_C];   // homogeneous solutions: y[-1] = 1, y[-2] = 0  /  y[-1] = 0, y[-2] = 1
    M2 pw[8];                                 // M_C^(2^d), d = 0..7: chunk-level scan inside a tile
};
struct ChainPlan {
    M2 mt;                                    // M_tile = M_C^256
    M2 pw[10];                                // (M_tile^per)^(2^d), d = 0..9: scan over the 1024 threads of k_biquad_chain
    int per;                                  // tiles per thread
};

__device__ __forceinline__ void mv(const M2 &m, double &p, double &q) {
    const double np = m.a * p + m.b * q, nq = m.c * p + m.d * q;
    p = np;
    q = nq;
}

// padded LDS index: lane stride C = 32 dwords would hit one bank; one pad word per 32
__device__ __forceinline__ int pidx(int i) { return i + (i >> 5); }

template <bool APPLY, bool VEC>
static __global__ __launch_bounds__(SP_IIR_WG) void k_biquad_tile(const float *__restrict__ x, int64_t n, BiquadPlan pl,
                                                                   const double *__restrict__ tile_in /*[ntiles][2]*/,
                                                                   double *__restrict__ tile_end /*[ntiles][2]*/,
                                                                   float *__restrict__ y) {
    __shared__ float xs[SP_IIR_TILE + SP_IIR_TILE / 32 + 8];
    __shared__ double st[2][SP_IIR_WG];
    const int64_t t0 = (int64_t)blockIdx.x * SP_IIR_TILE;
    const int tid = threadIdx.x;
    // tile (the two samples before each chunk are read from the tile or, for chunk 0, from global); 16-byte loads when
    // the tile is whole and the pointer allows
    const bool vec = VEC && t0 + SP_IIR_TILE <= n;
    if (vec) {
        const float4 *x4 = reinterpret_cast<const float4 *>(x + t0);
        float4 r[SP_IIR_TILE / 4 / SP_IIR_WG];
#pragma unroll
        for (int k = 0; k < SP_IIR_TILE / 4 / SP_IIR_WG; ++k) r[k] = x4[tid + SP_IIR_WG * k];
#pragma unroll
        for (int k = 0; k < SP_IIR_TILE / 4 / SP_IIR_WG; ++k) {
            const int i = 4 * (tid + SP_IIR_WG * k);
            xs[pidx(i)] = r[k].x;
            xs[pidx(i) + 1] = r[k].y;
            xs[pidx(i) + 2] = r[k].z;
            xs[pidx(i) + 3] = r[k].w;
        }
    } else {
#pragma unroll 4
        for (int i = tid; i < SP_IIR_TILE; i += SP_IIR_WG) {
            const int64_t g = t0 + i;
            xs[pidx(i)] = g < n ? x[g] : 0.f;
        }
    }
    __syncthreads();
    const int c0 = tid * SP_IIR_C;
    double xm1, xm2;
    if (tid == 0) {
        xm1 = t0 >= 1 ? (double)x[t0 - 1] : 0.0;
        xm2 = t0 >= 2 ? (double)x[t0 - 2] : 0.0;
    } else {
        xm1 = (double)xs[pidx(c0 - 1)];
        xm2 = (double)xs[pidx(c0 - 2)];
    }
    // the chunk from rest
    double yl[SP_IIR_C];
    double y1 = 0.0, y2 = 0.0;
#pragma unroll
    for (int j = 0; j < SP_IIR_C; ++j) {
        const double xc = (double)xs[pidx(c0 + j)];
        const double v = pl.b0 * xc + pl.b1 * xm1 + pl.b2 * xm2;
        const double yy = v - pl.a1 * y1 - pl.a2 * y2;
        yl[j] = yy;
        y2 = y1;
        y1 = yy;
        xm2 = xm1;
        xm1 = xc;
    }
    // inclusive scan of the chunk end states: f_i <- f_i + M^(2^d) f_{i - 2^d}
    double p = y1, q = y2;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        st[0][tid] = p;
        st[1][tid] = q;
        __syncthreads();
        if (tid >= (1 << d)) {
            double pp = st[0][tid - (1 << d)], qq = st[1][tid - (1 << d)];
            mv(pl.pw[d], pp, qq);
            p += pp;
            q += qq;
        }
        __syncthreads();
    }
    if (!APPLY) {
        if (tid == SP_IIR_WG - 1) {
            tile_end[2 * (int64_t)blockIdx.x] = p;
            tile_end[2 * (int64_t)blockIdx.x + 1] = q;
        }
        return;
    }
    // incoming state of this chunk: end state of the chunks before it (from rest) + M_C^tid applied to the tile's incoming
    // state -- the latter by running the tile state through the same scan slots: thread i needs M_C^i s_tile; i in binary
    st[0][tid] = p;
    st[1][tid] = q;
    __syncthreads();
    double sp = tid > 0 ? st[0][tid - 1] : 0.0, sq = tid > 0 ? st[1][tid - 1] : 0.0;
    double tp = tile_in[2 * (int64_t)blockIdx.x], tq = tile_in[2 * (int64_t)blockIdx.x + 1];
#pragma unroll
    for (int d = 0; d < 8; ++d)
        if (tid & (1 << d)) mv(pl.pw[d], tp, tq);
    sp += tp;
    sq += tq;
    __syncthreads();
    // outputs: rest response + homogeneous response to (sp, sq); staged through the (now free) tile for coalesced stores
#pragma unroll
    for (int j = 0; j < SP_IIR_C; ++j) xs[pidx(c0 + j)] = (float)(yl[j] + pl.alpha[j] * sp + pl.beta[j] * sq);
    __syncthreads();
    if (vec) {
        float4 *y4 = reinterpret_cast<float4 *>(y + t0);
#pragma unroll
        for (int k = 0; k < SP_IIR_TILE / 4 / SP_IIR_WG; ++k) {
            const int i = 4 * (tid + SP_IIR_WG * k);
            y4[tid + SP_IIR_WG * k] = make_float4(xs[pidx(i)], xs[pidx(i) + 1], xs[pidx(i) + 2], xs[pidx(i) + 3]);
        }
    } else {
#pragma unroll 4
        for (int i = tid; i < SP_IIR_TILE; i += SP_IIR_WG) {
            const int64_t g = t0 + i;
            if (g < n) y[g] = xs[pidx(i)];
        }
    }
}

// incoming state of every tile from the tiles' from-rest end states: s_{k+1} = M_tile s_k + f_k, s_0 = 0.  One workgroup of
// 1024 threads; thread i owns tiles [i per, (i+1) per).
static __global__ __launch_bounds__(1024) void k_biquad_chain(const double *__restrict__ tile_end, int64_t ntiles, ChainPlan cp,
                                                               double *__restrict__ tile_in) {
    __shared__ double st[2][1024];
    const int tid = threadIdx.x;
    const int64_t k0 = (int64_t)tid * cp.per;
    double p = 0.0, q = 0.0;
    for (int j = 0; j < cp.per; ++j) {
        const int64_t k = k0 + j;
        mv(cp.mt, p, q);                       // (tiles past the end contribute f = 0: the state just decays on)
        if (k < ntiles) {
            p += tile_end[2 * k];
            q += tile_end[2 * k + 1];
        }
    }
#pragma unroll
    for (int d = 0; d < 10; ++d) {
        st[0][tid] = p;
        st[1][tid] = q;
        __syncthreads();
        if (tid >= (1 << d)) {
            double pp = st[0][tid - (1 << d)], qq = st[1][tid - (1 << d)];
            mv(cp.pw[d], pp, qq);
            p += pp;
            q += qq;
        }
        __syncthreads();
    }
    st[0][tid] = p;
    st[1][tid] = q;
    __syncthreads();
    double sp = tid > 0 ? st[0][tid - 1] : 0.0, sq = tid > 0 ? st[1][tid - 1] : 0.0;
    for (int j = 0; j < cp.per; ++j) {
        const int64_t k = k0 + j;
        if (k >= ntiles) break;
        tile_in[2 * k] = sp;
        tile_in[2 * k + 1] = sq;
        mv(cp.mt, sp, sq);
        sp += tile_end[2 * k];
        sq += tile_end[2 * k + 1];
    }
}

static M2 mmul(const M2 &x, const M2 &y) {
    return M2{x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d};
}
static M2 mpow(M2 m, int64_t e) {
    M2 r{1, 0, 0, 1};
    while (e > 0) {
        if (e & 1) r = mmul(r, m);
        m = mmul(m, m);
        e >>= 1;
    }
    return r;
}

int64_t biquad_tiles(int64_t n) { return (n + SP_IIR_TILE - 1) / SP_IIR_TILE; }

// b[3], a[3] (a[0] != 0); x, y device float32 [n]; work: device scratch of 4 * biquad_tiles(n) doubles
int launch_biquad(LaunchCtx c, const double *b, const double *a, const float *x, int64_t n, float *y, double *work) {
    if (n < 1 || a[0] == 0.0) return -1;
    BiquadPlan pl;
    pl.b0 = b[0] / a[0];
    pl.b1 = b[1] / a[0];
    pl.b2 = b[2] / a[0];
    pl.a1 = a[1] / a[0];
    pl.a2 = a[2] / a[0];
    // homogeneous solutions and the one-chunk state map  (y[C-1], y[C-2]) = M_C (y[-1], y[-2])
    for (int k = 0; k < 2; ++k) {
        double y1 = k == 0 ? 1.0 : 0.0, y2 = k == 0 ? 0.0 : 1.0;
        for (int j = 0; j < SP_IIR_C; ++j) {
            const double yy = -pl.a1 * y1 - pl.a2 * y2;
            (k == 0 ? pl.alpha : pl.beta)[j] = yy;
            y2 = y1;
            y1 = yy;
        }
    }
    const M2 mc{pl.alpha[SP_IIR_C - 1], pl.beta[SP_IIR_C - 1], pl.alpha[SP_IIR_C - 2], pl.beta[SP_IIR_C - 2]};
    M2 m = mc;
    for (int d = 0; d < 8; ++d) {
        pl.pw[d] = m;
        m = mmul(m, m);
    }
    const int64_t nt = biquad_tiles(n);
    ChainPlan cp;
    cp.mt = m;                                            // M_C^256
    cp.per = (int)((nt + 1023) / 1024);
    M2 mp = mpow(cp.mt, cp.per);
    for (int d = 0; d < 10; ++d) {
        cp.pw[d] = mp;
        mp = mmul(mp, mp);
    }
    double *tile_end = work, *tile_in = work + 2 * nt;
    const bool vec = (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
    if (vec) hipLaunchKernelGGL((k_biquad_tile<false, true>), dim3((unsigned)nt), dim3(SP_IIR_WG), 0, c.stream, x, n, pl,
                                (const double *)nullptr, tile_end, (float *)nullptr);
    else hipLaunchKernelGGL((k_biquad_tile<false, false>), dim3((unsigned)nt), dim3(SP_IIR_WG), 0, c.stream, x, n, pl,
                            (const double *)nullptr, tile_end, (float *)nullptr);
    hipLaunchKernelGGL(k_biquad_chain, dim3(1), dim3(1024), 0, c.stream, (const double *)tile_end, nt, cp, tile_in);
    if (vec) hipLaunchKernelGGL((k_biquad_tile<true, true>), dim3((unsigned)nt), dim3(SP_IIR_WG), 0, c.stream, x, n, pl,
                                (const double *)tile_in, tile_end, y);
    else hipLaunchKernelGGL((k_biquad_tile<true, false>), dim3((unsigned)nt), dim3(SP_IIR_WG), 0, c.stream, x, n, pl,
                            (const double *)tile_in, tile_end, y);
    return 0;
}

}   // namespace sp
