// k_long.hip -- segments longer than one workgroup transform (the reference's default regime: Navr = 8 gives
// nwins = floor(nsig / 4.5), fft_analysis.py:2412-2418, e.g. 116 508 points for its own test_fftanal, :2950-2993).
// There are few such frames and each is large, so the path is a composition instead of one fused kernel:
//   k_long_pack      frames [f0, f0+m) of one signal: detrend (global record, or a per-frame record from
//                    k_long_segstats), window, -> complex rows S[m][nfft]   (fft_analysis.py:2148, :2156-2176, :362-388)
//   dev_fft_any      (spectral.hip) every row through the multi-pass FFT / Bluestein chirp-z, batched per launch
//   k_long_acc_*     sum over the rows of |X|^2 and Y conj(X) into float64 accumulators, one thread per bin, fixed
//                    order (deterministic)                                  (fft_analysis.py:391-393, :1946-1960, :444-446)
//   k_long_finish_*  sidedness, doubling, scale                             (fft_analysis.py:402-440, :2179-2203)
//   k_long_stft_out  or: the rows themselves as STFT output / power / centre-of-gravity moments
// Every sample is read once per chunk of frames that contains it; the spectra of a chunk make one round trip through HBM
// (they are at most SP_LONG_CHUNK_BYTES, so that round trip stays in the 256 MiB Infinity Cache).
#include "launch.h"
namespace sp {

// per-frame detrend record for the per-segment modes (matplotlib.mlab convention; fft_win(detrendwin=True), :2171):
// rec[b] = (a_re, a_im, s_re, s_im), the value removed at LOCAL index j of frame f0 + b is a + s j.
// mode 1: the frame's mean; mode 2: its least-squares line (slope = sum (j - jbar) x_j / (n (n^2 - 1) / 12)).
// One workgroup per frame, float64 sums.
template <bool CPLX>
static __global__ __launch_bounds__(1024) void k_long_segstats(const void *__restrict__ x, int64_t f0, int hop, int nfft,
                                                                int mode, float *__restrict__ rec) {
    __shared__ double sh[4][1024];
    const int64_t base = (f0 + blockIdx.x) * (int64_t)hop;
    const double jbar = 0.5 * (double)(nfft - 1);
    double s0 = 0, s1 = 0, u0 = 0, u1 = 0;
    for (int j = threadIdx.x; j < nfft; j += 1024) {
        const cf v = load_sample(x, base + j, CPLX);
        const double d = (double)j - jbar;
        s0 += (double)v.x;
        s1 += (double)v.y;
        u0 += d * (double)v.x;
        u1 += d * (double)v.y;
    }
    sh[0][threadIdx.x] = s0;
    sh[1][threadIdx.x] = s1;
    sh[2][threadIdx.x] = u0;
    sh[3][threadIdx.x] = u1;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
#pragma unroll
            for (int q = 0; q < 4; ++q) sh[q][threadIdx.x] += sh[q][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double n = (double)nfft;
        const double mr = sh[0][0] / n, mi = sh[1][0] / n;
        double sr = 0, si = 0;
        if (mode == 2 && nfft > 1) {
            const double den = n * (n * n - 1.0) / 12.0;
            sr = sh[2][0] / den;
            si = sh[3][0] / den;
        }
        float *r = rec + 4 * (int64_t)blockIdx.x;
        r[0] = (float)(mr - jbar * sr);
        r[1] = (float)(mi - jbar * si);
        r[2] = (float)sr;
        r[3] = (float)si;
    }
}

// S[b][j] = win[j] * (x[(f0+b) hop + j] - trend), j < nfft; blockIdx.y = b.  pseg (optional, zeroed by the caller):
// pseg[f0+b] += trapz |S[b][.]|^2 with unit spacing (fft_analysis.py:2174), one float64 atomic per workgroup.
template <bool CPLX>
static __global__ __launch_bounds__(256) void k_long_pack(const void *__restrict__ x, const float *__restrict__ win, int nfft,
                                                           int hop, int64_t f0, const float *__restrict__ trend, int lin,
                                                           const float *__restrict__ segrec, cf *__restrict__ S,
                                                           double *__restrict__ pseg) {
    __shared__ double sh[256];
    const int b = blockIdx.y;
    const int64_t base = (f0 + b) * (int64_t)hop;
    const cf tm = mk(trend[0], trend[1]), ts = mk(trend[2], trend[3]);
    cf ra = mk(0.f, 0.f), rs = mk(0.f, 0.f);
    if (segrec != nullptr) {
        ra = mk(segrec[4 * b], segrec[4 * b + 1]);
        rs = mk(segrec[4 * b + 2], segrec[4 * b + 3]);
    }
    cf *row = S + (int64_t)b * nfft;
    double pw = 0.0;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < nfft; j += gridDim.x * 256) {
        cf v = load_sample(x, base + j, CPLX);
        if (segrec != nullptr) v = mk(v.x - (ra.x + rs.x * (float)j), v.y - (ra.y + rs.y * (float)j));
        if (lin) {
            const float fi = (float)(base + j);
            v = mk(v.x - (tm.x + ts.x * fi), v.y - (tm.y + ts.y * fi));
        } else {
            v = v - tm;
        }
        v = win[j] * v;
        row[j] = v;
        pw += ((j == 0 || j == nfft - 1) ? 0.5 : 1.0) * (double)cnorm(v);
    }
    if (pseg != nullptr) {
        sh[threadIdx.x] = pw;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) atomicAdd(&pseg[f0 + b], sh[0]);
    }
}

// acc[k] += sum_b |S[b][k]|^2
static __global__ __launch_bounds__(256) void k_long_acc_psd(const cf *__restrict__ S, int64_t m, int nfft,
                                                              double *__restrict__ acc) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nfft) return;
    double a = acc[k];
    for (int64_t b = 0; b < m; ++b) a += (double)cnorm(S[b * nfft + k]);
    acc[k] = a;
}
// ayy[k] += sum_b |Y|^2,  axy[k] += sum_b Y conj(X)   (fft_analysis.py:393: Pxy = Y X*)
static __global__ __launch_bounds__(256) void k_long_acc_csd(const cf *__restrict__ Sx, const cf *__restrict__ Sy, int64_t m,
                                                              int nfft, double *__restrict__ ayy, double *__restrict__ axy) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nfft) return;
    double a = ayy[k], cr = axy[2 * (int64_t)k], ci = axy[2 * (int64_t)k + 1];
    for (int64_t b = 0; b < m; ++b) {
        const cf X = Sx[b * nfft + k], Y = Sy[b * nfft + k];
        a += (double)cnorm(Y);
        cr += (double)Y.x * (double)X.x + (double)Y.y * (double)X.y;
        ci += (double)Y.y * (double)X.x - (double)Y.x * (double)X.y;
    }
    ayy[k] = a;
    axy[2 * (int64_t)k] = cr;
    axy[2 * (int64_t)k + 1] = ci;
}
// out[slot(k)] = scale * doubling * acc[k]     (cplx: two doubles per bin)
static __global__ __launch_bounds__(256) void k_long_finish(const double *__restrict__ acc, int nfft, int sided, double scale,
                                                             int cplx, double *__restrict__ out) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nfft) return;
    const int slot = bin_slot(k, nfft, sided);
    if (slot < 0) return;
    const double m = scale * (bin_doubled(k, nfft, sided) ? 2.0 : 1.0);
    if (cplx) {
        out[2 * (int64_t)slot] = m * acc[2 * (int64_t)k];
        out[2 * (int64_t)slot + 1] = m * acc[2 * (int64_t)k + 1];
    } else {
        out[slot] = m * acc[k];
    }
}
// STFT rows: out[(f0+b) nb + slot] = amp (sqrt2 on doubled bins) X, or amp |X|^2  (same conventions as k_stft)
static __global__ __launch_bounds__(256) void k_long_stft_out(const cf *__restrict__ S, int nfft, int sided, float amp,
                                                               int out_power, void *__restrict__ out, int64_t f0, int nb) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nfft) return;
    const int slot = bin_slot(k, nfft, sided);
    if (slot < 0) return;
    const int64_t b = blockIdx.y;
    const cf v = S[b * nfft + k];
    const int64_t o = (f0 + b) * (int64_t)nb + slot;
    if (out_power) {
        reinterpret_cast<float *>(out)[o] = amp * cnorm(v);
    } else {
        const float a = bin_doubled(k, nfft, sided) ? amp * 1.41421356237309504880f : amp;
        reinterpret_cast<cf *>(out)[o] = a * v;
    }
}
// centre-of-gravity moments of row b (Doppler.py:43-58; same band rule as k_stft): acc[f0+b] = (sum ks |X|^2, sum |X|^2)
static __global__ __launch_bounds__(1024) void k_long_cog(const cf *__restrict__ S, int nfft, int klo, int khi,
                                                           cf *__restrict__ acc, int64_t f0) {
    __shared__ double sh[2][1024];
    const int64_t b = blockIdx.x;
    double num = 0, den = 0;
    for (int k = threadIdx.x; k < nfft; k += 1024) {
        const int ks = k < (nfft + 1) / 2 ? k : k - nfft;
        const int ka = ks < 0 ? -ks : ks;
        if (ka >= klo && ka <= khi) {
            const double p = (double)cnorm(S[b * nfft + k]);
            num += p * (double)ks;
            den += p;
        }
    }
    sh[0][threadIdx.x] = num;
    sh[1][threadIdx.x] = den;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) acc[f0 + b] = mk((float)sh[0][0], (float)sh[1][0]);
}

static unsigned bin_blocks(int nfft) { return (unsigned)((nfft + 255) / 256); }

int launch_long_segstats(LaunchCtx c, const void *x, bool cplx, int64_t f0, int64_t m, int hop, int nfft, int mode, float *rec) {
    if (m < 1) return -1;
    if (cplx) hipLaunchKernelGGL((k_long_segstats<true>), dim3((unsigned)m), dim3(1024), 0, c.stream, x, f0, hop, nfft, mode, rec);
    else hipLaunchKernelGGL((k_long_segstats<false>), dim3((unsigned)m), dim3(1024), 0, c.stream, x, f0, hop, nfft, mode, rec);
    return 0;
}
int launch_long_pack(LaunchCtx c, const void *x, bool cplx, const float *win, int nfft, int hop, int64_t f0, int64_t m,
                     const float *trend, bool lin, const float *segrec, cf *S, double *pseg) {
    if (m < 1 || m > 65535) return -1;
    // enough workgroups along the frame to fill the chip when there are few frames
    int64_t bx = (nfft + 255) / 256;
    const int64_t want = ((int64_t)c.ncu * 8 + m - 1) / m;
    if (bx > want) bx = want < 1 ? 1 : want;
    const dim3 grid((unsigned)bx, (unsigned)m);
    if (cplx) hipLaunchKernelGGL((k_long_pack<true>), grid, dim3(256), 0, c.stream, x, win, nfft, hop, f0, trend, lin ? 1 : 0, segrec, S, pseg);
    else hipLaunchKernelGGL((k_long_pack<false>), grid, dim3(256), 0, c.stream, x, win, nfft, hop, f0, trend, lin ? 1 : 0, segrec, S, pseg);
    return 0;
}
int launch_long_acc_psd(LaunchCtx c, const cf *S, int64_t m, int nfft, double *acc) {
    hipLaunchKernelGGL(k_long_acc_psd, dim3(bin_blocks(nfft)), dim3(256), 0, c.stream, S, m, nfft, acc);
    return 0;
}
int launch_long_acc_csd(LaunchCtx c, const cf *Sx, const cf *Sy, int64_t m, int nfft, double *ayy, double *axy) {
    hipLaunchKernelGGL(k_long_acc_csd, dim3(bin_blocks(nfft)), dim3(256), 0, c.stream, Sx, Sy, m, nfft, ayy, axy);
    return 0;
}
int launch_long_finish(LaunchCtx c, const double *acc, int nfft, int sided, double scale, bool cplx, double *out) {
    hipLaunchKernelGGL(k_long_finish, dim3(bin_blocks(nfft)), dim3(256), 0, c.stream, acc, nfft, sided, scale, cplx ? 1 : 0, out);
    return 0;
}
int launch_long_stft_out(LaunchCtx c, const cf *S, int64_t m, int nfft, int sided, float amp, int out_power, void *out,
                         int64_t f0, int nb) {
    if (m < 1 || m > 65535) return -1;
    hipLaunchKernelGGL(k_long_stft_out, dim3(bin_blocks(nfft), (unsigned)m), dim3(256), 0, c.stream, S, nfft, sided, amp,
                       out_power, out, f0, nb);
    return 0;
}
int launch_long_cog(LaunchCtx c, const cf *S, int64_t m, int nfft, int klo, int khi, cf *acc, int64_t f0) {
    if (m < 1) return -1;
    hipLaunchKernelGGL(k_long_cog, dim3((unsigned)m), dim3(1024), 0, c.stream, S, nfft, klo, khi, acc, f0);
    return 0;
}

}   // namespace sp
