// k_fft.hip -- batched C2C, Hilbert, FIR, cross-covariance launchers
#include "launch.h"
namespace sp {

int launch_fft_c2c(LaunchCtx c, const cf *in, cf *out, int64_t batch, int inverse, const Xf &xf, BigTw bt) {
    const int blocks = strided_blocks(xf.L, batch, c.ncu, xf.L == 4096 && !xf.blue ? 12 : 4);
#define M_(XT)                                                                                        \
    hipLaunchKernelGGL((k_fft_c2c<XT>), dim3(blocks), dim3(XT::C::WG), XT::C::lds_bytes(1), c.stream, in, out, batch, \
                       inverse, xf.tb, bt);
    SP_DISPATCH_X(xf, M_)
#undef M_
    return 0;
}

// one workgroup per FPW rows; the grid is padded to a multiple of 8 blocks (surplus blocks exit at once)
int launch_fft_strided(LaunchCtx c, const cf *in, cf *out, int64_t batch, int64_t in_rs, int64_t in_es, int64_t out_rs,
                       int64_t out_es, int conj_in, int conj_out, float scale, const Xf &xf, BigTw bt) {
    if (xf.blue) return -1;
    const int fpw = fpw_of(xf.L);
    if (batch % fpw) return -1;
    int64_t blocks = batch / fpw;
    blocks = (blocks + 7) / 8 * 8;
#define M_(XT)                                                                                        \
    hipLaunchKernelGGL((k_fft_strided<XT::L>), dim3((unsigned)blocks), dim3(XT::C::WG), XT::C::lds_bytes(1), c.stream, in, out, \
                       batch, in_rs, in_es, out_rs, out_es, conj_in, conj_out, scale, xf.tb, bt, 1);
    SP_DISPATCH_P(xf, M_)
#undef M_
    return 0;
}

static bool env_cols_noxpair() {
    const char *e = getenv("SP_COLS_NOXPAIR");     // A/B switch (read per call)
    return e && e[0] == '1';
}
static bool env_cols_nohalf() {
    const char *e = getenv("SP_COLS_NOHALF");      // A/B switch of the tests (read per call)
    return e && e[0] == '1';
}
// three-pass long transform pieces (power-of-two lengths; ncols is a multiple of the columns per workgroup)
int launch_fft_cols(LaunchCtx c, const cf *in, cf *out, int64_t ncols, int64_t nouter, int64_t es, int64_t os, int64_t twmul,
                    int conj_in, const Xf &xf, BigTw bt, int64_t hmask_n, ColsIn ci, int tw_outer) {
    if (xf.blue) return -1;
    int fpw = fpw_of(xf.L);
    if (ncols % fpw || nouter < 1) return -1;
    // first passes on real samples (kind 1 / 3), 256-point columns: 512-thread workgroups that own 32 adjacent columns (WM = 2,
    // SP_COLS_WIDE=1).  Measured SLOWER than 16 columns (round 3: 70.9 against 63.1 us for the Hilbert's first pass at 2^23 points,
    // 196.9 against 147.4 for the ccf's at 2^25): one workgroup per CU instead of two, and the twice as long row segments do
    // not pay for it.  Off by default.
    static const bool wide_ok = getenv("SP_COLS_WIDE") && getenv("SP_COLS_WIDE")[0] == '1';
    const bool wide = wide_ok && (ci.kind == 1 || ci.kind == 3) && xf.L == 256 && hmask_n == 0 && ncols % (2 * fpw) == 0;
    if (wide) fpw *= 2;
    const int64_t ncb = ncols / fpw, total = ncb * nouter;
    // (a multiple of the 2 or 3 workgroups a CU holds, so that the last round is a full one)
    const int64_t cap = (int64_t)c.ncu * (wide ? 3 : 6);       // several blocks per workgroup amortise its twiddle set-up
    const unsigned grid = (unsigned)(total < cap ? total : cap);
    if (wide) {
        using XT = XfPow2<256>;
        if (ci.kind == 1)
            hipLaunchKernelGGL((k_fft_cols<256, 1, false, 2>), dim3(grid), dim3(XT::C::WG * 2), XT::C::lds_bytes(1) * 2, c.stream, in, out, ncb,
                               nouter, es, os, twmul, conj_in, xf.tb, bt, hmask_n, ci, tw_outer);
        else
            hipLaunchKernelGGL((k_fft_cols<256, 3, false, 2>), dim3(grid), dim3(XT::C::WG * 2), XT::C::lds_bytes(1) * 2, c.stream, in, out, ncb,
                               nouter, es, os, twmul, conj_in, xf.tb, bt, hmask_n, ci, tw_outer);
        return 0;
    }
    // real samples in (kind 1 / 4: 64-byte pieces per row and array): pair neighbouring column blocks on one XCD
    if ((ci.kind == 1 || ci.kind == 4) && total % 16 == 0 && grid % 16 == 0 && !env_cols_noxpair()) tw_outer |= 2;
    // kind 1 whose samples end exactly at the middle row: the predicate-free form
    const bool half_exact = ci.kind == 1 && ci.r2 && ci.mom && nouter == 1 && xf.L >= 32 && ci.nreal == (int64_t)(xf.L / 2) * es &&
                            !env_cols_nohalf();
#define M_(XT)                                                                                        \
    if (half_exact) hipLaunchKernelGGL((k_fft_cols<XT::L, 4>), dim3(grid), dim3(XT::C::WG), XT::C::lds_bytes(1), c.stream, in, out, \
                                       ncb, nouter, es, os, twmul, conj_in, xf.tb, bt, hmask_n, ci, tw_outer);       \
    else if (ci.kind == 1) hipLaunchKernelGGL((k_fft_cols<XT::L, 1>), dim3(grid), dim3(XT::C::WG), XT::C::lds_bytes(1), c.stream, in, out, \
                                         ncb, nouter, es, os, twmul, conj_in, xf.tb, bt, hmask_n, ci, tw_outer);       \
    else if (ci.kind == 3) hipLaunchKernelGGL((k_fft_cols<XT::L, 3>), dim3(grid), dim3(XT::C::WG), XT::C::lds_bytes(1), c.stream, in, \
                                              out, ncb, nouter, es, os, twmul, conj_in, xf.tb, bt, hmask_n, ci, tw_outer); \
    else if (hmask_n > 0) hipLaunchKernelGGL((k_fft_cols<XT::L, 0, true>), dim3(grid), dim3(XT::C::WG), XT::C::lds_bytes(1), c.stream, \
                                             in, out, ncb, nouter, es, os, twmul, conj_in, xf.tb, bt, hmask_n, ci, tw_outer); \
    else hipLaunchKernelGGL((k_fft_cols<XT::L, 0>), dim3(grid), dim3(XT::C::WG), XT::C::lds_bytes(1), c.stream, in, out, ncb, nouter, \
                            es, os, twmul, conj_in, xf.tb, bt, hmask_n, ci, tw_outer);
    SP_DISPATCH_P(xf, M_)
#undef M_
    return 0;
}

int launch_fft_rows_rev(LaunchCtx c, const cf *in, cf *out, int64_t A, int64_t B, int conj_out, float scale, const Xf &xf,
                        RowsOut ro) {
    if (xf.blue) return -1;
    const int fpw = fpw_of(xf.L);
    if (A % fpw) return -1;
    const int64_t total = A * B / fpw, cap = (int64_t)c.ncu * 4;
    const unsigned grid = (unsigned)(total < cap ? total : cap);
#define M_(XT)                                                                                        \
    if (ro.co != nullptr && ro.kind == 2)                                                             \
        hipLaunchKernelGGL((k_fft_rows_rev<XT::L, 2>), dim3(grid), dim3(XT::C::WG), XT::C::lds_bytes(1), c.stream, in, out, A, B, \
                           conj_out, scale, xf.tb, ro);                                               \
    else if (ro.co != nullptr && ro.kind == 3)                                                        \
        hipLaunchKernelGGL((k_fft_rows_rev<XT::L, 3>), dim3(grid), dim3(XT::C::WG), XT::C::lds_bytes(1), c.stream, in, out, A, B, \
                           conj_out, scale, xf.tb, ro);                                               \
    else if (ro.co != nullptr)                                                                        \
        hipLaunchKernelGGL((k_fft_rows_rev<XT::L, 1>), dim3(grid), dim3(XT::C::WG), XT::C::lds_bytes(1), c.stream, in, out, A, B, \
                           conj_out, scale, xf.tb, ro);                                               \
    else hipLaunchKernelGGL((k_fft_rows_rev<XT::L, 0>), dim3(grid), dim3(XT::C::WG), XT::C::lds_bytes(1), c.stream, in, out, A, B, \
                            conj_out, scale, xf.tb, ro);
    SP_DISPATCH_P(xf, M_)
#undef M_
    return 0;
}

int launch_hilbert_rowsmid(LaunchCtx c, cf *Tm, int64_t A, int64_t B, const Xf &xc, BigTw btN, const cf *tw2c) {
    if (xc.blue || A < 2 || B < 2) return -1;
    const int64_t nslots = A * B / 2;
#define RM_(LL)                                                                                        \
    case LL: {                                                                                        \
        constexpr int HP = WgCfg<LL>::FPW / 2;                                                        \
        const int64_t iters = (nslots + HP - 1) / HP, cap = (int64_t)c.ncu * 4;                       \
        hipLaunchKernelGGL((k_hilbert_rowsmid<LL>), dim3((unsigned)(iters < cap ? iters : cap)), dim3(WgCfg<LL>::WG),   \
                           WgCfg<LL>::lds_bytes(1), c.stream, Tm, A, B, xc.tb, btN, tw2c);            \
        return 0;                                                                                     \
    }
    switch (xc.L) {
        RM_(32) RM_(64) RM_(128) RM_(256) RM_(512) RM_(1024) RM_(2048)
        default: return -1;
    }
#undef RM_
}

int launch_xc_rowsmid(LaunchCtx c, cf *Tm, int64_t A, int64_t B, const Xf &xc, const Xf &xc2, BigTw btL, BigTw btM) {
    if (xc.blue || xc2.blue || xc2.L * 2 != xc.L || A < 2 || B < 2) return -1;
    const int64_t nslots = A * B / 2;
#define XR_(LL)                                                                                        \
    case LL: {                                                                                        \
        constexpr int HP = WgCfg<LL>::FPW / 2;                                                        \
        const int64_t iters = (nslots + HP - 1) / HP, cap = (int64_t)c.ncu * 4;                       \
        const size_t lds = sizeof(cf) * (size_t)WgCfg<LL>::FPW * (LL + 32);                          \
        hipLaunchKernelGGL((k_xc_rowsmid<LL>), dim3((unsigned)(iters < cap ? iters : cap)), dim3(WgCfg<LL>::WG), lds, c.stream, Tm, A, B, \
                           xc.tb, xc2.tb, btL, btM);                                                  \
        return 0;                                                                                     \
    }
    switch (xc.L) {
        XR_(64) XR_(128) XR_(256) XR_(512) XR_(1024) XR_(2048)
        default: return -1;
    }
#undef XR_
}

int launch_fft_cols_lag(LaunchCtx c, const cf *in, int64_t ncols, int64_t nouter, int64_t es, int64_t os, const Xf &xf, RowsOut ro) {
    if (xf.blue) return -1;
    const int fpw = fpw_of(xf.L);
    if (ncols % fpw || nouter < 1) return -1;
    const int64_t ncb = ncols / fpw, total = ncb * nouter, cap = (int64_t)c.ncu * 6;
    const unsigned grid = (unsigned)(total < cap ? total : cap);
    if (total % 16 == 0 && grid % 16 == 0 && !env_cols_noxpair()) ro.kind |= 16;
#define CL_(LL)                                                                                        \
    case LL:                                                                                          \
        hipLaunchKernelGGL((k_fft_cols_lag<LL>), dim3(grid), dim3(WgCfg<LL>::WG), WgCfg<LL>::lds_bytes(1), c.stream, in, ncb, nouter, es, os, \
                           xf.tb, ro);                                                                \
        return 0;
    switch (xf.L) {
        CL_(64) CL_(128) CL_(256)
        default: return -1;
    }
#undef CL_
}

int launch_fft_cols_inv(LaunchCtx c, const cf *in, cf *out, int64_t ncols, int64_t nouter, int64_t es, int64_t os, int64_t twmul,
                        const Xf &xf, BigTw bt, float scale, const RowsOut *analytic) {
    if (xf.blue) return -1;
    const int fpw = fpw_of(xf.L);
    if (ncols % fpw || nouter < 1) return -1;
    const int64_t ncb = ncols / fpw, total = ncb * nouter, cap = (int64_t)c.ncu * 6;
    const unsigned grid = (unsigned)(total < cap ? total : cap);
    const RowsOut ro = analytic ? *analytic : RowsOut{nullptr, 0, 0, nullptr};
    // the whole row holds samples and pairs are 8-byte aligned: the predicate-free output form
    const bool full = analytic && ro.n == ro.Ltot && (((uintptr_t)ro.rx) & 7) == 0 && !env_cols_nohalf();
#define CI_(LL)                                                                                        \
    case LL:                                                                                          \
        if (analytic && full) hipLaunchKernelGGL((k_fft_cols_inv<LL, 3>), dim3(grid), dim3(WgCfg<LL>::WG), WgCfg<LL>::lds_bytes(1), c.stream, in, out, \
                                         ncb, nouter, es, os, twmul, xf.tb, bt, scale, ro);           \
        else if (analytic) hipLaunchKernelGGL((k_fft_cols_inv<LL, 2>), dim3(grid), dim3(WgCfg<LL>::WG), WgCfg<LL>::lds_bytes(1), c.stream, in, out, \
                                         ncb, nouter, es, os, twmul, xf.tb, bt, scale, ro);           \
        else hipLaunchKernelGGL((k_fft_cols_inv<LL, 0>), dim3(grid), dim3(WgCfg<LL>::WG), WgCfg<LL>::lds_bytes(1), c.stream, in, out, ncb, \
                                nouter, es, os, twmul, xf.tb, bt, scale, ro);                         \
        return 0;
    switch (xf.L) {
        CI_(64) CI_(128) CI_(256)
        default: return -1;
    }
#undef CI_
}

int launch_hilbert_mid(LaunchCtx c, cf *Z, int64_t M, BigTw bt) {
    const int64_t n = M / 2 + 1;
    const int64_t cap = (int64_t)c.ncu * 16;
    const int64_t b = (n + 255) / 256;
    hipLaunchKernelGGL(k_hilbert_mid, dim3((unsigned)(b < cap ? b : cap)), dim3(256), 0, c.stream, Z, M, bt);
    return 0;
}

int launch_hilbert(LaunchCtx c, const float *x, int64_t n_in, int64_t x_ld, int64_t batch, const Xf &xf, cf *out,
                   const cf *H) {
    const int blocks = strided_blocks(xf.L, batch, c.ncu, xf.L == 4096 && !xf.blue && !H ? 3 : 4);
#define M_(XT)                                                                                        \
    if (H) hipLaunchKernelGGL((k_hilbert<XT, true>), dim3(blocks), dim3(XT::C::WG), XT::C::lds_bytes(1), c.stream, x, n_in, \
                              x_ld, batch, xf.tb, out, H);                                            \
    else hipLaunchKernelGGL((k_hilbert<XT, false>), dim3(blocks), dim3(XT::C::WG), XT::C::lds_bytes(1), c.stream, x, n_in, \
                            x_ld, batch, xf.tb, out, H);
    SP_DISPATCH_X(xf, M_)
#undef M_
    return 0;
}

int launch_fftfilt(LaunchCtx c, const float *x, int64_t n, int ntaps, const cf *Hs, const Xf &xf, float *y) {
    const int64_t Lb = xf.L - (ntaps - 1);
    const int64_t npairs = ((n + Lb - 1) / Lb + 1) / 2;
    // interior pairs: both blocks and their outputs wholly inside [0, n): pair p reads [2pLb-(P-1), 2pLb+Lb+N-(P-1))
    int64_t pi0 = 1, pi1 = 0;
    if (npairs >= 3) {
        // largest p with 2 p Lb - (P-1) + Lb + N <= n
        pi1 = (n - xf.L - Lb + (ntaps - 1)) / (2 * Lb) + 1;
        if (pi1 > npairs) pi1 = npairs;
        if (pi1 < pi0) pi1 = pi0;
    } else {
        pi0 = pi1 = 0;
    }
#define M_(XT)                                                                                        \
    if (pi1 > pi0) {                                                                                  \
        const int blocks = strided_blocks(xf.L, pi1 - pi0, c.ncu, 6);      /* (three resident per CU) */ \
        hipLaunchKernelGGL((k_fftfilt<XT::L, false>), dim3(blocks), dim3(XT::C::WG), XT::C::lds_bytes(1), c.stream, x, n, \
                           ntaps, Hs, xf.tb, y, pi0, pi1);                                            \
    }                                                                                                 \
    if (pi0 > 0 || pi1 <= pi0) {                                                                      \
        const int64_t e1 = pi1 > pi0 ? pi0 : npairs;                                                  \
        hipLaunchKernelGGL((k_fftfilt<XT::L, true>), dim3(strided_blocks(xf.L, e1, c.ncu)), dim3(XT::C::WG),     \
                           XT::C::lds_bytes(1), c.stream, x, n, ntaps, Hs, xf.tb, y, (int64_t)0, e1);   \
    }                                                                                                 \
    if (pi1 > pi0 && pi1 < npairs) {                                                                  \
        hipLaunchKernelGGL((k_fftfilt<XT::L, true>), dim3(strided_blocks(xf.L, npairs - pi1, c.ncu)), dim3(XT::C::WG), \
                           XT::C::lds_bytes(1), c.stream, x, n, ntaps, Hs, xf.tb, y, pi1, npairs);      \
    }
    SP_DISPATCH_P(xf, M_)
#undef M_
    return 0;
}

int launch_xcorr(LaunchCtx c, const float *x1, const float *x2, int64_t n, const double *mom, const Xf &xf, float *co) {
#define M_(XT)                                                                                        \
    hipLaunchKernelGGL((k_xcorr<XT::L>), dim3(1), dim3(XT::C::WG), XT::C::lds_bytes(1), c.stream, x1, x2, n, mom, \
                       xf.tb, co);
    SP_DISPATCH_P(xf, M_)
#undef M_
    return 0;
}

int launch_moments(LaunchCtx c, const void *x, bool cplx, int64_t n, int mode, double *partial, double *out_d,
                   float *trend_f, int nsignals, int64_t x_cs) {
    int64_t nb = (n + 256 * 16 - 1) / (256 * 16);
    const int64_t cap = nsignals > 1 ? (4096 / nsignals < 8 ? 8 : 4096 / nsignals) : 4096;   // partial scratch: 4096 records
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    if ((int64_t)nsignals * nb > 4096) return -1;
    const bool lin = mode == 2;
    const dim3 grid((unsigned)nb, (unsigned)nsignals);
    if (cplx) {
        if (lin) hipLaunchKernelGGL((k_moments_partial<true, true>), grid, dim3(256), 0, c.stream, x, n, partial, x_cs);
        else hipLaunchKernelGGL((k_moments_partial<true, false>), grid, dim3(256), 0, c.stream, x, n, partial, x_cs);
    } else {
        if (lin) hipLaunchKernelGGL((k_moments_partial<false, true>), grid, dim3(256), 0, c.stream, x, n, partial, x_cs);
        else hipLaunchKernelGGL((k_moments_partial<false, false>), grid, dim3(256), 0, c.stream, x, n, partial, x_cs);
    }
    hipLaunchKernelGGL(k_moments_finish, dim3(nsignals), dim3(256), 0, c.stream, partial, (int)nb, n, mode, out_d, trend_f);
    return 0;
}

// ccf: means / sums of squares of two real signals (x_cs samples apart) and the normalisation record xc_out[4] in two launches
int launch_moments_xc(LaunchCtx c, const float *x, int64_t n, double *partial, double *out_d, double *xc_out, int64_t x_cs) {
    int64_t nb = (n + 256 * 16 - 1) / (256 * 16);
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL((k_moments_partial<false, false>), dim3((unsigned)nb, 2), dim3(256), 0, c.stream, x, n, partial, x_cs);
    hipLaunchKernelGGL(k_moments_finish_xc, dim3(1), dim3(256), 0, c.stream, partial, (int)nb, n, out_d, xc_out);
    return 0;
}

int launch_transpose(LaunchCtx c, const void *in, void *out, int64_t rows, int64_t cols, int elem_bytes) {
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    if (elem_bytes == 4)
        hipLaunchKernelGGL((k_transpose<float>), grid, dim3(32, 8), 0, c.stream, (const float *)in, (float *)out, rows, cols);
    else if (elem_bytes == 8)
        hipLaunchKernelGGL((k_transpose<cf>), grid, dim3(32, 8), 0, c.stream, (const cf *)in, (cf *)out, rows, cols);
    else
        return -1;
    return 0;
}

static int ew_blocks(int64_t n, int ncu) {
    int64_t b = (n + 255) / 256;
    const int64_t cap = (int64_t)ncu * 16;
    return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

int launch_transpose_c(LaunchCtx c, const cf *in, cf *out, int64_t rows, int64_t cols, int conj, float scale, int64_t batch) {
    if (batch < 1 || batch > 65535 || (rows + 31) / 32 > 65535) return -1;
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32), (unsigned)batch);
    hipLaunchKernelGGL(k_transpose_c, grid, dim3(32, 8), 0, c.stream, in, out, rows, cols, conj, scale);
    return 0;
}
int launch_pack_real(LaunchCtx c, const float *x, int64_t n_in, const double *mean, int64_t L, cf *out) {
    hipLaunchKernelGGL(k_pack_real, dim3(ew_blocks(L, c.ncu)), dim3(256), 0, c.stream, x, n_in, mean, L, out);
    return 0;
}
int launch_cmul_vec(LaunchCtx c, const cf *a, const cf *b, int64_t n, int conj_out, cf *out, int64_t batch) {
    if (batch < 1 || batch > 65535) return -1;
    hipLaunchKernelGGL(k_cmul_vec, dim3(ew_blocks(n, c.ncu), (unsigned)batch), dim3(256), 0, c.stream, a, b, n, conj_out, out);
    return 0;
}
int launch_blue_pre(LaunchCtx c, const cf *in, const cf *chirp, int64_t n, int64_t L, int conj_in, cf *out, int64_t batch) {
    if (batch < 1 || batch > 65535) return -1;
    hipLaunchKernelGGL(k_blue_pre, dim3(ew_blocks(L, c.ncu), (unsigned)batch), dim3(256), 0, c.stream, in, chirp, n, L, conj_in, out);
    return 0;
}
int launch_blue_post(LaunchCtx c, const cf *in, const cf *chirp, int64_t n, int conj_out, float scale, cf *out, int64_t batch,
                     int64_t in_ld) {
    if (batch < 1 || batch > 65535) return -1;
    hipLaunchKernelGGL(k_blue_post, dim3(ew_blocks(n, c.ncu), (unsigned)batch), dim3(256), 0, c.stream, in, chirp, n, conj_out, scale,
                       out, in_ld);
    return 0;
}
int launch_hilbert_mask(LaunchCtx c, cf *X, int64_t n) {
    hipLaunchKernelGGL(k_hilbert_mask, dim3(ew_blocks(n, c.ncu)), dim3(256), 0, c.stream, X, n);
    return 0;
}
int frame_sum_slices(int ncu, int nch, int nfft, int64_t nframes) {
    const int nb = (nfft + 255) / 256;
    int64_t slices = (8 * (int64_t)ncu + (int64_t)nb * nch - 1) / ((int64_t)nb * nch);
    if (slices > (nframes + 7) / 8) slices = (nframes + 7) / 8;
    if (slices < 1) slices = 1;
    if (slices > 65535) slices = 65535;
    return (int)slices;
}
// part: scratch of frame_sum_slices(...) * nch * nfft * 2 doubles
int launch_frame_sum(LaunchCtx c, const void *x, bool cplx, int64_t x_ld, int nch, int nfft, int hop, int64_t nframes,
                     const float *trend, bool lin, double *out, double *part) {
    // enough frame slices to fill the chip, each at least 8 frames long
    const int nb = (nfft + 255) / 256;
    const int slices = frame_sum_slices(c.ncu, nch, nfft, nframes);
    const dim3 grid(nb, (unsigned)slices, nch);
    if (lin) hipLaunchKernelGGL((k_frame_sum<true>), grid, dim3(256), 0, c.stream, x, cplx ? 1 : 0, x_ld, nfft, hop, nframes, trend, part);
    else hipLaunchKernelGGL((k_frame_sum<false>), grid, dim3(256), 0, c.stream, x, cplx ? 1 : 0, x_ld, nfft, hop, nframes, trend, part);
    const int64_t count = (int64_t)nch * nfft * 2;
    hipLaunchKernelGGL(k_frame_sum_reduce, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c.stream, (const double *)part, slices,
                       count, out);
    return 0;
}
int launch_spec_mul(LaunchCtx c, cf *X, const cf *H, int64_t n) {
    hipLaunchKernelGGL(k_spec_mul, dim3(ew_blocks(n, c.ncu)), dim3(256), 0, c.stream, X, H, n);
    return 0;
}
int launch_xc_pack(LaunchCtx c, const float *x1, const float *x2, int64_t n, int64_t L, const double *mom, cf *z) {
    hipLaunchKernelGGL(k_xc_pack, dim3(ew_blocks(L, c.ncu)), dim3(256), 0, c.stream, x1, x2, n, L, mom, z);
    return 0;
}
int launch_xc_mid_half(LaunchCtx c, const cf *Z, int64_t L, BigTw bt, cf *Zp) {
    const int64_t b = (L / 2 + 255) / 256, cap = (int64_t)c.ncu * 16;
    hipLaunchKernelGGL(k_xc_mid_half, dim3((unsigned)(b < cap ? b : cap)), dim3(256), 0, c.stream, Z, L, bt, Zp);
    return 0;
}
int launch_xc_mid(LaunchCtx c, const cf *Z, int64_t L, cf *R) {
    hipLaunchKernelGGL(k_xc_mid, dim3(ew_blocks(L, c.ncu)), dim3(256), 0, c.stream, Z, L, R);
    return 0;
}
int launch_xc_out(LaunchCtx c, const cf *r, int64_t n, int64_t L, const double *mom, float *co) {
    hipLaunchKernelGGL(k_xc_out, dim3(ew_blocks(2 * n - 1, c.ncu)), dim3(256), 0, c.stream, r, n, L, mom, co);
    return 0;
}

}   // namespace sp
