// k_epilogue.hip -- the fft_pwelch epilogue on device-resident averaged spectra (fft_analysis.py:489-648,
// :1662-1688): coherence, mean-squared coherence, cross-phase, linear amplitude spectra, and the auto-/cross-correlations
// Rxx / Ryy / Rxy / iCxy by inverse FFT of the (re-symmetrised) spectra, corrcoef = Rxy / sqrt(Ex Ey).
// Inputs are what sp_welch_csd leaves on the device: pxx[nb], pyy[nch][nb], pxy[nch][nb] (float64, complex interleaved).
// Elementwise algebra in float64; the length-nfft inverse transforms run through the complex64 FFT (any length).
#include "launch.h"
namespace sp {

// Cxy = Pxy / sqrt(|Pxx||Pyy|), Cxy2 = |Pxy|^2 / (|Pxx||Pyy|), phi = atan2(Im, Re), L = sqrt(|ENBW P|) (x sqrt2 on the
// doubled bins of a one-sided spectrum: [1:-1], plus the last bin when nfft is odd; :526-540)
static __global__ __launch_bounds__(256) void k_epi_elem(const double *__restrict__ pxx, const double *__restrict__ pyy,
                                                          const double *__restrict__ pxy, int nch, int nb, int nfft, int onesided,
                                                          double enbw, double *__restrict__ cxy, double *__restrict__ cxy2,
                                                          double *__restrict__ phi, double *__restrict__ lxx,
                                                          double *__restrict__ lyy, double *__restrict__ lxy) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (k >= nb) return;
    const bool dbl = onesided && k >= 1 && (k <= nb - 2 || ((nfft & 1) && k == nb - 1));
    const double amp = dbl ? 1.41421356237309504880 : 1.0;
    const double xx = pxx[k], yy = pyy[(int64_t)c * nb + k];
    const double re = pxy[2 * ((int64_t)c * nb + k)], im = pxy[2 * ((int64_t)c * nb + k) + 1];
    const double den = fabs(xx) * fabs(yy), rs = 1.0 / sqrt(den);
    const int64_t o = (int64_t)c * nb + k;
    cxy[2 * o] = re * rs;
    cxy[2 * o + 1] = im * rs;
    cxy2[o] = (re * re + im * im) / den;
    phi[o] = atan2(im, re);
    if (c == 0) lxx[k] = amp * sqrt(fabs(enbw * xx));
    lyy[o] = amp * sqrt(fabs(enbw * yy));
    lxy[o] = amp * sqrt(enbw * sqrt(re * re + im * im));
}

// spectrum of signal s laid out for the inverse transform, complex64 [nsig][nfft]:
//   s = 0: Pxx;  1..nch: Pyy_c;  nch+1..2nch: Pxy_c;  2nch+1..3nch: Cxy_c
// one-sided input (bins [0, nb)): [1:-1] halved (and the last bin when nfft is odd), bins beyond nb zero, then the
// Hermitian mirror with real DC / Nyquist -- numpy.fft.irfft semantics (:544-577); two-sided input (shifted): ifftshift.
// The inverse transforms run in float32: every row is divided by a scale of its own first and multiplied back in float64 behind
// the transform (spectra of a volts-scale 1e-10 signal, P ~ 1e-26, lost their halved bins to float32 denormals; large
// ones overflowed).  rowmax[s] = max_k of Pxx (s = 0) / Pyy_c (s = 1 + c); scale of Pxx: rowmax[0], Pyy_c: rowmax[1 + c],
// Pxy_c: sqrt(rowmax[0] rowmax[1 + c]) >= |Pxy_c[k]| (Cauchy-Schwarz bin by bin), the coherence: 1.
static __global__ __launch_bounds__(256) void k_epi_rowmax(const double *__restrict__ pxx, const double *__restrict__ pyy, int nb,
                                                            double *__restrict__ rowmax) {
    __shared__ double sh[256];
    const int s = blockIdx.x;
    const double *p = s == 0 ? pxx : pyy + (int64_t)(s - 1) * nb;
    double m = 0.0;
    for (int k = threadIdx.x; k < nb; k += 256) m = fmax(m, fabs(p[k]));
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) rowmax[s] = sh[0];
}
__device__ __forceinline__ double epi_row_scale(const double *__restrict__ rowmax, int s, int nch) {
    double v = 1.0;
    if (s == 0) v = rowmax[0];
    else if (s <= nch) v = rowmax[s];
    else if (s <= 2 * nch) v = sqrt(rowmax[0] * rowmax[s - nch]);
    return v > 0.0 && isfinite(v) ? v : 1.0;
}

static __global__ __launch_bounds__(256) void k_epi_spec(const double *__restrict__ pxx, const double *__restrict__ pyy,
                                                          const double *__restrict__ pxy, const double *__restrict__ cxy, int nch,
                                                          int nb, int nfft, int onesided, cf *__restrict__ X,
                                                          const double *__restrict__ rowmax) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int s = blockIdx.y;
    if (j >= nfft) return;
    const double inv = 1.0 / epi_row_scale(rowmax, s, nch);
    auto value = [&](int k, double &re, double &im) __attribute__((always_inline)) {
        if (s == 0) {
            re = pxx[k];
            im = 0.0;
        } else if (s <= nch) {
            re = pyy[(int64_t)(s - 1) * nb + k];
            im = 0.0;
        } else if (s <= 2 * nch) {
            re = pxy[2 * ((int64_t)(s - 1 - nch) * nb + k)];
            im = pxy[2 * ((int64_t)(s - 1 - nch) * nb + k) + 1];
        } else {
            re = cxy[2 * ((int64_t)(s - 1 - 2 * nch) * nb + k)];
            im = cxy[2 * ((int64_t)(s - 1 - 2 * nch) * nb + k) + 1];
        }
    };
    double re = 0.0, im = 0.0;
    if (onesided) {
        const int nh = nfft / 2 + 1;
        const bool upper = j >= nh;
        const int k = upper ? nfft - j : j;                    // mirror source
        if (k < nb) {
            value(k, re, im);
            const bool iscoh = s > 2 * nch;                      // iCxy: the coherence is transformed as it stands (:575)
            if (!iscoh && k >= 1 && (k <= nb - 2 || ((nfft & 1) && k == nb - 1))) {
                re *= 0.5;
                im *= 0.5;
            }
            if (k == 0 || ((nfft & 1) == 0 && k == nfft / 2)) im = 0.0;
            if (upper) im = -im;
        }
    } else {
        const int half = nfft / 2;                               // ifftshift: X[j] = P[(j + half) % nfft]
        int k = j + half;
        if (k >= nfft) k -= nfft;
        value(k, re, im);
    }
    X[(int64_t)s * nfft + j] = mk((float)(re * inv), (float)(im * inv));
}

// R[s][fftshift slot] = sqrt(nfft) * x[s][j] as float64 pairs (real part only for one-sided input: irfft is real);
// the zero-lag values E[s] = sqrt(nfft) x[s][0] of Pxx / Pyy_c go to ee[1 + nch][2]
static __global__ __launch_bounds__(256) void k_epi_corr(const cf *__restrict__ X, int nch, int nfft, int onesided,
                                                          double *__restrict__ rxx, double *__restrict__ ryy,
                                                          double *__restrict__ rxy, double *__restrict__ icxy,
                                                          double *__restrict__ ee, const double *__restrict__ rowmax) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int s = blockIdx.y;
    if (j >= nfft) return;
    const double sc = sqrt((double)nfft) * epi_row_scale(rowmax, s, nch);
    const cf v = X[(int64_t)s * nfft + j];
    const double re = sc * (double)v.x, im = onesided ? 0.0 : sc * (double)v.y;
    int slot = j + nfft / 2;
    if (slot >= nfft) slot -= nfft;
    double *dst;
    if (s == 0) dst = rxx;
    else if (s <= nch) dst = ryy + 2 * (int64_t)(s - 1) * nfft;
    else if (s <= 2 * nch) dst = rxy + 2 * (int64_t)(s - 1 - nch) * nfft;
    else dst = icxy + 2 * (int64_t)(s - 1 - 2 * nch) * nfft;
    dst[2 * (int64_t)slot] = re;
    dst[2 * (int64_t)slot + 1] = im;
    if (j == 0 && s <= nch) {
        ee[2 * s] = re;
        ee[2 * s + 1] = im;
    }
}

// corrcoef_c = Rxy_c / sqrt(Ex Ey_c)  (complex square root of the product, like numpy's for complex input; :590-596)
static __global__ __launch_bounds__(256) void k_epi_corrcoef(const double *__restrict__ rxy, const double *__restrict__ ee, int nch,
                                                              int nfft, double *__restrict__ cc) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (j >= nfft) return;
    const double ar = ee[0], ai = ee[1], br = ee[2 * (c + 1)], bi = ee[2 * (c + 1) + 1];
    const double pr = ar * br - ai * bi, pi = ar * bi + ai * br;
    // principal square root of p
    const double m = sqrt(pr * pr + pi * pi);
    double sr = sqrt(0.5 * (m + pr)), si = sqrt(fmax(0.0, 0.5 * (m - pr)));
    if (pi < 0.0) si = -si;
    const double d = sr * sr + si * si;
    const double xr = rxy[2 * ((int64_t)c * nfft + j)], xi = rxy[2 * ((int64_t)c * nfft + j) + 1];
    cc[2 * ((int64_t)c * nfft + j)] = (xr * sr + xi * si) / d;
    cc[2 * ((int64_t)c * nfft + j) + 1] = (xi * sr - xr * si) / d;
}

static unsigned blocks_of(int n) { return (unsigned)((n + 255) / 256); }

int launch_epi_elem(LaunchCtx c, const double *pxx, const double *pyy, const double *pxy, int nch, int nb, int nfft, int onesided,
                    double enbw, double *cxy, double *cxy2, double *phi, double *lxx, double *lyy, double *lxy) {
    if (nch < 1 || nch > 65535) return -1;
    hipLaunchKernelGGL(k_epi_elem, dim3(blocks_of(nb), (unsigned)nch), dim3(256), 0, c.stream, pxx, pyy, pxy, nch, nb, nfft, onesided,
                       enbw, cxy, cxy2, phi, lxx, lyy, lxy);
    return 0;
}
int launch_epi_spec(LaunchCtx c, const double *pxx, const double *pyy, const double *pxy, const double *cxy, int nch, int nb,
                    int nfft, int onesided, cf *X, double *rowmax) {
    if (3 * nch + 1 > 65535) return -1;
    hipLaunchKernelGGL(k_epi_rowmax, dim3((unsigned)(nch + 1)), dim3(256), 0, c.stream, pxx, pyy, nb, rowmax);
    hipLaunchKernelGGL(k_epi_spec, dim3(blocks_of(nfft), (unsigned)(3 * nch + 1)), dim3(256), 0, c.stream, pxx, pyy, pxy, cxy, nch, nb,
                       nfft, onesided, X, rowmax);
    return 0;
}
int launch_epi_corr(LaunchCtx c, const cf *X, int nch, int nfft, int onesided, double *rxx, double *ryy, double *rxy, double *icxy,
                    double *ee, double *cc, const double *rowmax) {
    hipLaunchKernelGGL(k_epi_corr, dim3(blocks_of(nfft), (unsigned)(3 * nch + 1)), dim3(256), 0, c.stream, X, nch, nfft, onesided, rxx,
                       ryy, rxy, icxy, ee, rowmax);
    hipLaunchKernelGGL(k_epi_corrcoef, dim3(blocks_of(nfft), (unsigned)nch), dim3(256), 0, c.stream, (const double *)rxy,
                       (const double *)ee, nch, nfft, cc);
    return 0;
}

}   // namespace sp
