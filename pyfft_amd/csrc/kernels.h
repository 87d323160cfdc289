// kernels.h -- gfx950 kernels of the spectral hot path, all built on WgFft (fft_core.h).
//
// Layout in HBM: signals are contiguous sample vectors (float32 or interleaved complex64);
// frames are never materialised -- frame g is the window [g*hop, g*hop+N) of the signal.
// A workgroup hosts FPW = 256/T transform groups (T = N/16 threads each); a group owns a run
// of consecutive frames (Welch/STFT) or a strided set of rows/blocks (FFT/Hilbert/FIR).
#pragma once
#include "fft_core.h"
#include <stdint.h>

namespace sp {

template <int N> struct WgCfg {
    using PL = FftPlan<N>;
    static constexpr int R = PL::R, T = PL::T;
    static constexpr int WG = T >= 256 ? T : 256;
    static constexpr int FPW = WG / T;
    static constexpr int LDS_PER = PL::LDS_ELEMS;              // complex elements per transform image
    static constexpr size_t lds_bytes(int nbuf) { return (size_t)FPW * LDS_PER * sizeof(cf) * nbuf; }
};

enum { SIDED_ONE = 1, SIDED_TWO = 2, SIDED_RAW = 3 };

// bin k of an N-point spectrum -> output slot and amplitude/power weights for a sidedness
// (fft_analysis.py:2179-2193 / :402-428).  returns -1 when the bin is dropped.
template <int N> __device__ __forceinline__ int bin_slot(int k, int sided) {
    if (sided == SIDED_ONE) return k < N / 2 ? k : -1;
    if (sided == SIDED_TWO) return (k + N / 2) & (N - 1);
    return k;
}
template <int N> __device__ __forceinline__ bool bin_doubled(int k, int sided) {
    return sided == SIDED_ONE && k >= 1 && k <= N / 2 - 2;      // [1:-1] of the cropped array (Q1)
}

__device__ __forceinline__ cf load_sample(const void *x, int64_t i, bool cplx) {
    if (cplx) return reinterpret_cast<const cf *>(x)[i];
    return mk(reinterpret_cast<const float *>(x)[i], 0.f);
}

// ------------------------------------------------------------------------------------------
// A7  batched C2C FFT (fft_analysis.py:2096-2116).  inverse through conj(fft(conj(.)))/N.
// ------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(WgCfg<N>::WG) void k_fft_c2c(const cf *__restrict__ in, cf *__restrict__ out,
                                                           int64_t batch, int inverse, const cf *__restrict__ twt) {
    using C = WgCfg<N>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *smem = reinterpret_cast<cf *>(smem_raw);
    const int grp = threadIdx.x / C::T, tid = threadIdx.x % C::T;
    cf *lds = smem + grp * C::LDS_PER;
    WgFft<N, false> f;
    f.load_twiddles(twt, tid);
    const float sgn = inverse ? -1.f : 1.f;
    const float scl = inverse ? 1.f / N : 1.f;
    const int64_t stride = (int64_t)gridDim.x * C::FPW;
    for (int64_t b0 = (int64_t)blockIdx.x * C::FPW; b0 < batch; b0 += stride) {
        const int64_t b = b0 + grp;
        const bool act = b < batch;
        const int64_t bl = act ? b : batch - 1;          // clamped: loads stay unconditional
        cf v[C::R];
#pragma unroll
        for (int t = 0; t < C::R; ++t) v[t] = in[bl * N + tid + C::T * t];
#pragma unroll
        for (int t = 0; t < C::R; ++t) v[t] = mk(v[t].x, sgn * v[t].y);
        f.template run<true>(v, lds, lds, tid);
        if (act) {
#pragma unroll
            for (int t = 0; t < C::R; ++t) out[b * N + tid + C::T * t] = mk(scl * v[t].x, sgn * scl * v[t].y);
        }
    }
}

// ------------------------------------------------------------------------------------------
// A3+A4  fused Welch PSD (fft_analysis.py:2156-2176 loop + :1946 |X|^2 + :1980 mean).
// Each group owns frames [gid*fpg, (gid+1)*fpg); |X|^2 is accumulated in registers over the
// run, one partial spectrum per group goes to HBM.  mean[2] (device) is subtracted before
// the window (global detrend :2148).
// ------------------------------------------------------------------------------------------
template <int N, bool CPLX>
__global__ __launch_bounds__(WgCfg<N>::WG) void k_welch(const void *__restrict__ x, const float *__restrict__ win,
                                                         int hop, int64_t nframes, int64_t fpg,
                                                         const float *__restrict__ mean, const cf *__restrict__ twt,
                                                         float *__restrict__ partial) {
    using C = WgCfg<N>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *smem = reinterpret_cast<cf *>(smem_raw);
    const int grp = threadIdx.x / C::T, tid = threadIdx.x % C::T;
    cf *lds = smem + grp * C::LDS_PER;
    WgFft<N, false> f;
    f.load_twiddles(twt, tid);
    float w[C::R], acc[C::R];
#pragma unroll
    for (int t = 0; t < C::R; ++t) {
        w[t] = win[tid + C::T * t];
        acc[t] = 0.f;
    }
    const cf mu = mk(mean[0], mean[1]);
    const int64_t gid = (int64_t)blockIdx.x * C::FPW + grp;
    const int64_t g0 = gid * fpg;
    for (int64_t i = 0; i < fpg; ++i) {
        const int64_t g = g0 + i;
        // frames past the end are clamped to the last one and weighted 0: every load is unconditional
        // (a per-load predicate makes hipcc branch around each load and serialise the round trips)
        const float keep = g < nframes ? 1.f : 0.f;
        const int64_t base = (g < nframes ? g : nframes - 1) * hop + tid;
        cf v[C::R];
#pragma unroll
        for (int t = 0; t < C::R; ++t) v[t] = load_sample(x, base + C::T * t, CPLX);
#pragma unroll
        for (int t = 0; t < C::R; ++t) v[t] = w[t] * (v[t] - mu);
        f.template run<true>(v, lds, lds, tid);
#pragma unroll
        for (int t = 0; t < C::R; ++t) acc[t] += keep * cnorm(v[t]);
    }
#pragma unroll
    for (int t = 0; t < C::R; ++t) partial[gid * N + tid + C::T * t] = acc[t];
}

// sum partial[G][N] over G in double, apply sidedness + scale -> out[nbins] (double).
// block = 64 bins x 16 slices of the group range (1024 threads); deterministic order.
#define SP_FIN_BINS 64
#define SP_FIN_SLICES 16
template <int N>
__global__ __launch_bounds__(SP_FIN_BINS *SP_FIN_SLICES) void k_welch_finish(const float *__restrict__ partial, int64_t G,
                                                                              int sided, double scale,
                                                                              double *__restrict__ out) {
    __shared__ double sh[SP_FIN_SLICES][SP_FIN_BINS];
    const int lane = threadIdx.x % SP_FIN_BINS, sl = threadIdx.x / SP_FIN_BINS;
    const int k = blockIdx.x * SP_FIN_BINS + lane;
    double s = 0.0;
    if (k < N)
        for (int64_t g = sl; g < G; g += SP_FIN_SLICES) s += (double)partial[g * N + k];
    sh[sl][lane] = s;
    __syncthreads();
    if (sl == 0 && k < N) {
        const int slot = bin_slot<N>(k, sided);
        if (slot >= 0) {
            double tot = 0.0;
#pragma unroll
            for (int j = 0; j < SP_FIN_SLICES; ++j) tot += sh[j][lane];
            out[slot] = tot * scale * (bin_doubled<N>(k, sided) ? 2.0 : 1.0);
        }
    }
}

// ------------------------------------------------------------------------------------------
// A5  fft_pwelch core (fft_analysis.py:362-393): reference x against channel y_c.
// grid.y = channel.  partial layout per (channel, group): [4][N] = |X|^2, |Y|^2, Re, Im of Y conj(X)
// ------------------------------------------------------------------------------------------
template <int N, bool CPLX>
__global__ __launch_bounds__(WgCfg<N>::WG) void k_welch_csd(const void *__restrict__ x, const void *__restrict__ y,
                                                             int64_t y_ld, const float *__restrict__ win, int hop,
                                                             int64_t nframes, int64_t fpg,
                                                             const float *__restrict__ mean_x,
                                                             const float *__restrict__ mean_y,
                                                             const cf *__restrict__ twt, float *__restrict__ partial,
                                                             int64_t groups_total) {
    using C = WgCfg<N>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *smem = reinterpret_cast<cf *>(smem_raw);
    const int grp = threadIdx.x / C::T, tid = threadIdx.x % C::T;
    cf *lds = smem + grp * C::LDS_PER;
    const int ch = blockIdx.y;
    WgFft<N, false> f;
    f.load_twiddles(twt, tid);
    float w[C::R], axx[C::R], ayy[C::R];
    cf axy[C::R];
#pragma unroll
    for (int t = 0; t < C::R; ++t) {
        w[t] = win[tid + C::T * t];
        axx[t] = ayy[t] = 0.f;
        axy[t] = mk(0.f, 0.f);
    }
    const cf mux = mk(mean_x[0], mean_x[1]);
    const cf muy = mk(mean_y[2 * ch], mean_y[2 * ch + 1]);
    const int64_t yoff = (int64_t)ch * y_ld;
    const int64_t gid = (int64_t)blockIdx.x * C::FPW + grp;
    const int64_t g0 = gid * fpg;
    for (int64_t i = 0; i < fpg; ++i) {
        const int64_t g = g0 + i;
        const float keep = g < nframes ? 1.f : 0.f;
        const int64_t base = (g < nframes ? g : nframes - 1) * hop + tid;
        cf vx[C::R], vy[C::R];
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            vx[t] = load_sample(x, base + C::T * t, CPLX);
            vy[t] = load_sample(y, yoff + base + C::T * t, CPLX);
        }
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            vx[t] = w[t] * (vx[t] - mux);
            vy[t] = w[t] * (vy[t] - muy);
        }
        f.template run<true>(vx, lds, lds, tid);
        f.template run<true>(vy, lds, lds, tid);
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            axx[t] += keep * cnorm(vx[t]);
            ayy[t] += keep * cnorm(vy[t]);
            cf p = cmulc(vy[t], vx[t]);     // Y conj(X)  (fft_analysis.py:393)
            axy[t] = axy[t] + keep * p;
        }
    }
    float *p = partial + ((int64_t)ch * groups_total + gid) * 4 * N;
#pragma unroll
    for (int t = 0; t < C::R; ++t) {
        const int k = tid + C::T * t;
        p[k] = axx[t];
        p[N + k] = ayy[t];
        p[2 * N + k] = axy[t].x;
        p[3 * N + k] = axy[t].y;
    }
}

// out layouts: pxx[nbins] (from channel 0's copy), pyy[nch][nbins], pxy[nch][nbins][2]
template <int N>
__global__ __launch_bounds__(SP_FIN_BINS *SP_FIN_SLICES) void k_csd_finish(const float *__restrict__ partial, int64_t G,
                                                                            int nch, int sided, double scale,
                                                                            double *__restrict__ pxx,
                                                                            double *__restrict__ pyy,
                                                                            double *__restrict__ pxy) {
    __shared__ double sh[4][SP_FIN_SLICES][SP_FIN_BINS];
    const int lane = threadIdx.x % SP_FIN_BINS, sl = threadIdx.x / SP_FIN_BINS;
    const int k = blockIdx.x * SP_FIN_BINS + lane;
    const int ch = blockIdx.y;
    const int nb = sided == SIDED_ONE ? N / 2 : N;
    double s[4] = {0, 0, 0, 0};
    const float *p = partial + (int64_t)ch * G * 4 * N;
    if (k < N)
        for (int64_t g = sl; g < G; g += SP_FIN_SLICES) {
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += (double)p[(g * 4 + j) * N + k];
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) sh[j][sl][lane] = s[j];
    __syncthreads();
    if (sl == 0 && k < N) {
        const int slot = bin_slot<N>(k, sided);
        if (slot >= 0) {
            double tot[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < SP_FIN_SLICES; ++q) tot[j] += sh[j][q][lane];
            const double m = scale * (bin_doubled<N>(k, sided) ? 2.0 : 1.0);
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
// pseg (optional): trapz of |win*(x-mean)|^2 over the frame, unit spacing (:2174).
// ------------------------------------------------------------------------------------------
template <int N, bool CPLX>
__global__ __launch_bounds__(WgCfg<N>::WG) void k_stft(const void *__restrict__ x, const float *__restrict__ win,
                                                        int hop, int64_t nframes, int64_t fpg,
                                                        const float *__restrict__ mean, const cf *__restrict__ twt,
                                                        int sided, float amp, int out_power, void *__restrict__ out,
                                                        double *__restrict__ pseg) {
    using C = WgCfg<N>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *smem = reinterpret_cast<cf *>(smem_raw);
    const int grp = threadIdx.x / C::T, tid = threadIdx.x % C::T;
    cf *lds = smem + grp * C::LDS_PER;
    WgFft<N, false> f;
    f.load_twiddles(twt, tid);
    float w[C::R];
#pragma unroll
    for (int t = 0; t < C::R; ++t) w[t] = win[tid + C::T * t];
    const cf mu = mk(mean[0], mean[1]);
    const int nb = sided == SIDED_ONE ? N / 2 : N;
    const int64_t gid = (int64_t)blockIdx.x * C::FPW + grp;
    const int64_t g0 = gid * fpg;
    for (int64_t i = 0; i < fpg; ++i) {
        const int64_t g = g0 + i;
        const bool act = g < nframes;
        const int64_t base = (act ? g : nframes - 1) * hop + tid;
        cf v[C::R];
        float pw = 0.f;
#pragma unroll
        for (int t = 0; t < C::R; ++t) v[t] = load_sample(x, base + C::T * t, CPLX);
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            v[t] = w[t] * (v[t] - mu);
            const int n = tid + C::T * t;
            pw += ((n == 0 || n == N - 1) ? 0.5f : 1.f) * cnorm(v[t]);
        }
        if (pseg != nullptr && act) atomicAdd(&pseg[g], (double)pw);
        f.template run<true>(v, lds, lds, tid);
        if (act) {
#pragma unroll
            for (int t = 0; t < C::R; ++t) {
                const int k = tid + C::T * t;
                const int slot = bin_slot<N>(k, sided);
                if (slot < 0) continue;
                if (out_power) {
                    reinterpret_cast<float *>(out)[g * nb + slot] = amp * cnorm(v[t]);
                } else {
                    const float a = bin_doubled<N>(k, sided) ? amp * 1.41421356237309504880f : amp;
                    reinterpret_cast<cf *>(out)[g * nb + slot] = a * v[t];
                }
            }
        }
    }
}

// tiled transpose [rows][cols] -> [cols][rows], elements of ESZ bytes (4 or 8) through LDS
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

// ------------------------------------------------------------------------------------------
// A10  analytic signal (hilbert.py:54-67): fft -> zero [nyq+1:], double [1:nyq) -> ifft, in one
// workgroup per row (even N here; odd / long lengths go through the generic path).
// ------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(WgCfg<N>::WG) void k_hilbert(const float *__restrict__ x, int64_t n_in, int64_t x_ld,
                                                           int64_t batch, const cf *__restrict__ twt,
                                                           cf *__restrict__ out) {
    using C = WgCfg<N>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *smem = reinterpret_cast<cf *>(smem_raw);
    const int grp = threadIdx.x / C::T, tid = threadIdx.x % C::T;
    cf *lds = smem + grp * C::LDS_PER;
    WgFft<N, false> f;
    f.load_twiddles(twt, tid);
    constexpr int nyq = N / 2;
    const int64_t stride = (int64_t)gridDim.x * C::FPW;
    for (int64_t b0 = (int64_t)blockIdx.x * C::FPW; b0 < batch; b0 += stride) {
        const int64_t b = b0 + grp;
        const bool act = b < batch;
        const int64_t bl = act ? b : batch - 1;
        cf v[C::R];
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int n = tid + C::T * t;
            const float a = x[bl * x_ld + (n < n_in ? n : n_in - 1)];     // clamped, unconditional
            v[t] = mk(n < n_in ? a : 0.f, 0.f);
        }
        f.template run<true>(v, lds, lds, tid);
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int k = tid + C::T * t;
            const float h = (k == 0 || k == nyq) ? 1.f : (k < nyq ? 2.f : 0.f);
            v[t] = mk(h * v[t].x, -h * v[t].y);          // mask, then conj for the inverse
        }
        f.template run<true>(v, lds, lds, tid);
        if (act) {
#pragma unroll
            for (int t = 0; t < C::R; ++t) out[b * N + tid + C::T * t] = mk(v[t].x * (1.f / N), -v[t].y * (1.f / N));
        }
    }
}

// ------------------------------------------------------------------------------------------
// F1  causal FIR by overlap-save; two real blocks ride in one complex transform (h real).
// Block b yields y[b*L : (b+1)*L) from x[b*L-(P-1) : b*L+L), L = N-(P-1).
// Hs[k] = FFT_N(h)[k] / N  (scale of the inverse folded in).
// ------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(WgCfg<N>::WG) void k_fftfilt(const float *__restrict__ x, int64_t n, int ntaps,
                                                           const cf *__restrict__ Hs, const cf *__restrict__ twt,
                                                           float *__restrict__ y) {
    using C = WgCfg<N>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *smem = reinterpret_cast<cf *>(smem_raw);
    const int grp = threadIdx.x / C::T, tid = threadIdx.x % C::T;
    cf *lds = smem + grp * C::LDS_PER;
    WgFft<N, false> f;
    f.load_twiddles(twt, tid);
    const int P1 = ntaps - 1;
    const int64_t L = N - P1;
    const int64_t nblocks = (n + L - 1) / L;
    const int64_t npairs = (nblocks + 1) / 2;
    cf H[C::R];
#pragma unroll
    for (int t = 0; t < C::R; ++t) H[t] = Hs[tid + C::T * t];
    const int64_t stride = (int64_t)gridDim.x * C::FPW;
    for (int64_t p0 = (int64_t)blockIdx.x * C::FPW; p0 < npairs; p0 += stride) {
        const int64_t p = p0 + grp;
        const bool act = p < npairs;
        const int64_t s0 = 2 * p * L - P1, s1 = s0 + L;      // first input sample of each block
        cf v[C::R];
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int m = tid + C::T * t;
            const int64_t i0 = s0 + m, i1 = s1 + m;
            const bool in0 = act && i0 >= 0 && i0 < n, in1 = act && i1 >= 0 && i1 < n;
            const float a = x[in0 ? i0 : 0], b = x[in1 ? i1 : 0];      // clamped, unconditional
            v[t] = mk(in0 ? a : 0.f, in1 ? b : 0.f);
        }
        f.template run<true>(v, lds, lds, tid);
#pragma unroll
        for (int t = 0; t < C::R; ++t) v[t] = cconj(cmul(v[t], H[t]));
        f.template run<true>(v, lds, lds, tid);
        if (act) {
#pragma unroll
            for (int t = 0; t < C::R; ++t) {
                const int m = tid + C::T * t;
                if (m >= P1) {
                    const int64_t o0 = s0 + m, o1 = s1 + m;
                    if (o0 < n) y[o0] = v[t].x;
                    if (o1 < n) y[o1] = -v[t].y;        // conj of the inverse trick
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// A11  cross-covariance at all lags for n <= N/2 (ccf.py:74-76) in one workgroup:
// z = (x1-m1) + i (x2-m2) zero-padded to N;  A conj(B) = Im(Z[k] Z[N-k])/2 + i (|Z[k]|^2-|Z[N-k]|^2)/4
// moments[0..3] = mean1, mean2, 1/(n*std1*std2), unused  (device)
// ------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(WgCfg<N>::WG) void k_xcorr(const float *__restrict__ x1, const float *__restrict__ x2,
                                                         int64_t n, const double *__restrict__ mom,
                                                         const cf *__restrict__ twt, float *__restrict__ co) {
    using C = WgCfg<N>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *smem = reinterpret_cast<cf *>(smem_raw);
    const int grp = threadIdx.x / C::T, tid = threadIdx.x % C::T;
    // only group 0 carries data; the others transform zeros so every thread meets the same barriers
    cf *lds = smem + grp * C::LDS_PER;
    WgFft<N, false> f;
    f.load_twiddles(twt, tid);
    const bool act = grp == 0;
    const float m1 = (float)mom[0], m2 = (float)mom[1];
    const float nrm = (float)mom[2];
    cf v[C::R];
#pragma unroll
    for (int t = 0; t < C::R; ++t) {
        const int i = tid + C::T * t;
        const bool in = act && i < n;
        const float a = x1[in ? i : 0], b = x2[in ? i : 0];
        v[t] = in ? mk(a - m1, b - m2) : mk(0.f, 0.f);
    }
    f.template run<true>(v, lds, lds, tid);
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
    f.template run<true>(v, lds, lds, tid);
    if (act) {
#pragma unroll
        for (int t = 0; t < C::R; ++t) {
            const int i = tid + C::T * t;          // r[i] = v.x / N   (imag ~ 0)
            const float r = v[t].x * (nrm / N);
            // 'full' order: j = lag + (n-1);  lag >= 0 -> r[lag], lag < 0 -> r[N+lag]
            if (i < n) co[(n - 1) + i] = r;
            else if (i > N - n) co[i - (N - n + 1)] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------
// reductions: sum / sum of squares in double
// ------------------------------------------------------------------------------------------
// partial[block][4] = sum re, sum im, sum re^2+im^2, 0
template <bool CPLX>
__global__ void k_moments_partial(const void *__restrict__ x, int64_t n, double *__restrict__ partial) {
    double s0 = 0, s1 = 0, s2 = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const cf a = load_sample(x, i, CPLX);
        s0 += a.x;
        s1 += a.y;
        s2 += (double)a.x * a.x + (double)a.y * a.y;
    }
    __shared__ double sh[3][256];
    sh[0][threadIdx.x] = s0;
    sh[1][threadIdx.x] = s1;
    sh[2][threadIdx.x] = s2;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
            sh[2][threadIdx.x] += sh[2][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 4 + 0] = sh[0][0];
        partial[blockIdx.x * 4 + 1] = sh[1][0];
        partial[blockIdx.x * 4 + 2] = sh[2][0];
    }
}

// one block of 256: out_d[0..1] = mean (double), out_d[2] = sum|x|^2, out_f[0..1] = mean (float)
__global__ void k_moments_finish(const double *__restrict__ partial, int nblocks, int64_t n,
                                 double *__restrict__ out_d, float *__restrict__ out_f) {
    __shared__ double sh[3][256];
    double s0 = 0, s1 = 0, s2 = 0;
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
        s0 += partial[b * 4];
        s1 += partial[b * 4 + 1];
        s2 += partial[b * 4 + 2];
    }
    sh[0][threadIdx.x] = s0;
    sh[1][threadIdx.x] = s1;
    sh[2][threadIdx.x] = s2;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
            sh[2][threadIdx.x] += sh[2][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out_d[0] = sh[0][0] / (double)n;
        out_d[1] = sh[1][0] / (double)n;
        out_d[2] = sh[2][0];
        if (out_f) {
            out_f[0] = (float)(sh[0][0] / (double)n);
            out_f[1] = (float)(sh[1][0] / (double)n);
        }
    }
}

}   // namespace sp
