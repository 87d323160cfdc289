// k_csd.hip -- reference-vs-channels cross-spectral density launchers
#include "launch.h"
namespace sp {

int launch_csd(LaunchCtx c, const void *x, const void *y, bool cplx, int nch, int64_t y_ld, const float *win, int hop,
               int64_t nframes, const float *trend_x, const float *trend_y, bool lin, const Xf &xf, float *partial,
               const RunPart &rp, int segmean) {
#define L_(XT, CP, LN)                                                                                \
    hipLaunchKernelGGL((k_welch_csd<XT, CP, LN>), dim3(rp.blocks, nch), dim3(XT::C::WG), XT::C::lds_bytes(1), c.stream, x, \
                       y, y_ld, win, hop, nframes, rp.fpg, trend_x, trend_y, xf.tb, partial, rp.groups, segmean)
#define M_(XT)                                                                                        \
    if (cplx) {                                                                                       \
        if (lin) L_(XT, true, true);                                                                  \
        else L_(XT, true, false);                                                                     \
    } else {                                                                                          \
        if (lin) L_(XT, false, true);                                                                 \
        else L_(XT, false, false);                                                                    \
    }
    SP_DISPATCH_X(xf, M_)
#undef M_
#undef L_
    return 0;
}

int launch_csd_finish(LaunchCtx c, const float *partial, int64_t G, const Xf &xf, int nch, int sided, double scale,
                      double *pxx, double *pyy, double *pxy) {
    const int n = xf.tb.n;
    hipLaunchKernelGGL(k_csd_finish, dim3((n + SP_FIN_BINS - 1) / SP_FIN_BINS, nch), dim3(SP_FIN_BINS * SP_FIN_SLICES), 0,
                       c.stream, partial, G, xf.L, n, nch, sided, scale, pxx, pyy, pxy);
    return 0;
}

// real x, y, power-of-two n >= 32: one transform per (frame, channel)
bool csd_rp_eligible(const Xf &xf) { return !xf.blue && xf.L >= 32 && xf.L <= 8192; }

int launch_csd_rp(LaunchCtx c, const float *x, const float *y, int nch, int64_t y_ld, const float *win, int hop,
                  int64_t nframes, const float *trend_x, const float *trend_y, bool lin, const Xf &xf, float *partial,
                  const RunPart &rp) {
#define RP_(NN)                                                                                       \
    case NN:                                                                                          \
        if (lin) hipLaunchKernelGGL((k_welch_csd_rp<NN, true>), dim3(rp.blocks, nch), dim3(WgCfg<NN>::WG),       \
                                    WgCfg<NN>::lds_bytes(1), c.stream, x, y, y_ld, win, hop, nframes, rp.fpg, trend_x,  \
                                    trend_y, xf.tb, partial, rp.groups);                              \
        else hipLaunchKernelGGL((k_welch_csd_rp<NN, false>), dim3(rp.blocks, nch), dim3(WgCfg<NN>::WG),          \
                                WgCfg<NN>::lds_bytes(1), c.stream, x, y, y_ld, win, hop, nframes, rp.fpg, trend_x,      \
                                trend_y, xf.tb, partial, rp.groups);                                  \
        break;
    switch (xf.L) {
        RP_(32) RP_(64) RP_(128) RP_(256) RP_(512) RP_(1024) RP_(2048) RP_(4096) RP_(8192)
        default: return -1;
    }
#undef RP_
    return 0;
}

// reference-once form: Zx = packed pair spectra of x, then one transform per (channel, frame pair)
int launch_pairspec(LaunchCtx c, const float *x, const float *win, int hop, int64_t nframes, const float *trend, bool lin,
                    const Xf &xf, const RunPart &rp, cf *Zx) {
#define PS_(NN)                                                                                       \
    case NN:                                                                                          \
        if (lin) hipLaunchKernelGGL((k_pairspec<NN, true>), dim3(rp.blocks), dim3(WgCfg<NN>::WG), WgCfg<NN>::lds_bytes(1),    \
                                    c.stream, x, win, hop, nframes, rp.fpg, trend, xf.tb, Zx);         \
        else hipLaunchKernelGGL((k_pairspec<NN, false>), dim3(rp.blocks), dim3(WgCfg<NN>::WG), WgCfg<NN>::lds_bytes(1),       \
                                c.stream, x, win, hop, nframes, rp.fpg, trend, xf.tb, Zx);             \
        break;
    switch (xf.L) {
        PS_(32) PS_(64) PS_(128) PS_(256) PS_(512) PS_(1024) PS_(2048) PS_(4096) PS_(8192)
        default: return -1;
    }
#undef PS_
    return 0;
}

// workgroups of the selected k_welch_csd_pair instantiation one CU keeps resident (0: unknown)
int csd_pair_resident(const Xf &xf, bool lin, bool onepass) {
    if (onepass) return resident_per_cu((const void *)k_welch_csd_pair<4096, false, true>, WgCfg<4096>::WG, WgCfg<4096>::lds_bytes(1));
#define CR_(NN)                                                                                       \
    case NN:                                                                                          \
        return lin ? resident_per_cu((const void *)k_welch_csd_pair<NN, true>, WgCfg<NN>::WG, WgCfg<NN>::lds_bytes(1)) \
                   : resident_per_cu((const void *)k_welch_csd_pair<NN, false>, WgCfg<NN>::WG, WgCfg<NN>::lds_bytes(1));
    switch (xf.L) {
        CR_(32) CR_(64) CR_(128) CR_(256) CR_(512) CR_(1024) CR_(2048) CR_(4096) CR_(8192)
        default: return 0;
    }
#undef CR_
}

int launch_csd_pair(LaunchCtx c, const float *y, int nch, int64_t y_ld, const float *win, int hop, int64_t nframes,
                    float *trend_y, bool lin, const Xf &xf, const cf *Zx, float *partial, const RunPart &rp, cf *spartial) {
    const int cf_ = (rp.blocks <= 65535 && !getenv("SP_CSD_RUNFAST")) ? 1 : 0;       // channel-fastest block order (grid.y <= 65535)
    const dim3 grid_ = cf_ ? dim3(nch, rp.blocks) : dim3(rp.blocks, nch);
    if (spartial) {
        // one-pass mean detrend (see the kernel): nfft 4096 at hop 2048 only
        if (lin || xf.L != 4096 || hop != 2048) return -1;
        hipLaunchKernelGGL((k_welch_csd_pair<4096, false, true>), grid_, dim3(WgCfg<4096>::WG), WgCfg<4096>::lds_bytes(1), c.stream, y,
                           y_ld, win, hop, nframes, rp.fpg, trend_y, xf.tb, Zx, partial, rp.groups, cf_, spartial);
        return 0;
    }
#define CP_(NN)                                                                                       \
    case NN:                                                                                          \
        if (lin) hipLaunchKernelGGL((k_welch_csd_pair<NN, true>), grid_, dim3(WgCfg<NN>::WG),         \
                                    WgCfg<NN>::lds_bytes(1), c.stream, y, y_ld, win, hop, nframes, rp.fpg, trend_y, xf.tb, Zx, \
                                    partial, rp.groups, cf_, (cf *)nullptr);                          \
        else hipLaunchKernelGGL((k_welch_csd_pair<NN, false>), grid_, dim3(WgCfg<NN>::WG),            \
                                WgCfg<NN>::lds_bytes(1), c.stream, y, y_ld, win, hop, nframes, rp.fpg, trend_y, xf.tb, Zx,     \
                                partial, rp.groups, cf_, (cf *)nullptr);                              \
        break;
    switch (xf.L) {
        CP_(32) CP_(64) CP_(128) CP_(256) CP_(512) CP_(1024) CP_(2048) CP_(4096) CP_(8192)
        default: return -1;
    }
#undef CP_
    return 0;
}

int launch_csd_pair_finish(LaunchCtx c, const float *partial, int64_t G, const Xf &xf, int nch, int sided, double scale,
                           double *pyy, double *pxy, const double *st_y, const double *st_x, const cf *Wf, const float *trend_x,
                           const float *trend_y, int64_t nmean, int64_t M) {
    const int n = xf.tb.n;
    hipLaunchKernelGGL(k_csd_pair_finish, dim3((n + SP_FIN_BINS - 1) / SP_FIN_BINS, nch), dim3(SP_FIN_BINS * SP_FIN_SLICES), 0,
                       c.stream, partial, G, n, nch, sided, scale, pyy, pxy, st_y, st_x, Wf, trend_x, trend_y, nmean, M);
    return 0;
}
// block sums of one real signal in SP_COLSUM_SLICES slices (out: [slices][H] complex), see k_colsum_real
int launch_colsum_real(LaunchCtx c, const float *x, const float *trend, int H, int64_t M, cf *out) {
    hipLaunchKernelGGL(k_colsum_real, dim3((H + 255) / 256, SP_COLSUM_SLICES), dim3(256), 0, c.stream, x, trend, H, M, out);
    return 0;
}

int launch_csd_rp_finish(LaunchCtx c, const float *partial, int64_t G, const Xf &xf, int nch, int sided, double scale,
                         double *pxx, double *pyy, double *pxy) {
    const int n = xf.tb.n;
    hipLaunchKernelGGL(k_csd_rp_finish, dim3((n + SP_FIN_BINS - 1) / SP_FIN_BINS, nch), dim3(SP_FIN_BINS * SP_FIN_SLICES), 0,
                       c.stream, partial, G, n, nch, sided, scale, pxx, pyy, pxy);
    return 0;
}

int launch_csdm_transpose(LaunchCtx c, const cf *Xs, cf *Xt, int nch, int64_t mc, int nb) {
    dim3 grid((unsigned)((nb + 31) / 32), (unsigned)((mc + 31) / 32), (unsigned)nch);
    hipLaunchKernelGGL(k_csdm_transpose, grid, dim3(32, 8), 0, c.stream, Xs, Xt, nch, mc, nb);
    return 0;
}

int launch_csdm_gemm(LaunchCtx c, const cf *Xt, int nch, int64_t mc, int nb, double *G) {
    const int nblk = (nch + SP_CM_B - 1) / SP_CM_B;
    // frame slices: enough workgroups (>= 8 per resident slot) that the last partial round costs little
    const int64_t wgs = (int64_t)nb * (nblk * (nblk + 1) / 2);
    int slices = (int)(((int64_t)c.ncu * 4 * 8 + wgs - 1) / wgs);
    const int max_slices = (int)((mc + 4 * SP_CM_F - 1) / (4 * SP_CM_F));        // at least 128 frames per slice
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    int64_t fs = (mc + slices - 1) / slices;
    fs = (fs + SP_CM_F - 1) / SP_CM_F * SP_CM_F;
    slices = (int)((mc + fs - 1) / fs);
    hipLaunchKernelGGL(k_csdm_gemm, dim3(nb, nblk * nblk, slices), dim3(256), 0, c.stream, Xt, nch, mc, nblk, G, fs);
    return 0;
}

int launch_csdm_finish(LaunchCtx c, double *G, int nch, int nb, double scale, int blk) {
    const int64_t total = (int64_t)nb * nch * nch;
    int64_t b = (total + 255) / 256;
    if (b > (int64_t)c.ncu * 16) b = (int64_t)c.ncu * 16;
    hipLaunchKernelGGL(k_csdm_finish, dim3((int)b), dim3(256), 0, c.stream, G, nch, nb, scale, blk);
    hipLaunchKernelGGL(k_csdm_mirror, dim3((int)b), dim3(256), 0, c.stream, G, nch, nb, blk);
    return 0;
}

// MFMA path: Xs[c][g][k] -> Xt2[k][g][c] (zero padded), then one workgroup per (bin, superblock pair, frame slice)
int launch_csdm_transpose_kgc(LaunchCtx c, const cf *Xs, cf *Xt, int nch, int nchp, int64_t m, int64_t mp, int nb) {
    dim3 grid((unsigned)((nb + 31) / 32), (unsigned)(nchp / 32), (unsigned)mp);
    hipLaunchKernelGGL(k_csdm_transpose_kgc, grid, dim3(32, 8), 0, c.stream, Xs, Xt, nch, nchp, m, mp, nb);
    return 0;
}

int launch_csdm_mfma(LaunchCtx c, const cf *Xt, int nch, int nchp, int64_t mp, int nb, double *G) {
    const int nsb = nchp / 64;
    const int npair = nsb * (nsb - 1) / 2;
    // Work units = (bin, frame slice), all of equal cost, 2 resident workgroups per CU (204 VGPRs).  Slices (multiples
    // of 256 frames) only when there are too few bins for >= 4 rounds; then the units that would form a partial last
    // round (e.g. the 2049th bin of cfg5: 2048 = 4 full rounds) are launched separately, cut 8 times finer, so that no
    // CU waits for a straggler.  Sliced units add into G atomically.
    const int64_t slots = (int64_t)c.ncu * 2;
    const int64_t per_bin = nsb + npair;                                         // workgroups per unit (grid.y)
    int slices = (int)((slots * 4 + (int64_t)nb * per_bin - 1) / ((int64_t)nb * per_bin));
    const int max_slices = (int)((mp + 255) / 256);
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    int64_t fs = (mp + slices - 1) / slices;
    fs = (fs + 255) / 256 * 256;
    slices = (int)((mp + fs - 1) / fs);
    const int64_t units = (int64_t)nb * slices;
    int64_t main_units = units * per_bin >= slots ? (units * per_bin / slots) * slots / per_bin : units;
    if (fs < 8 * 32) main_units = units;                                          // cannot cut finer
    auto go = [&](int64_t u0, int64_t n, int sl, int64_t f, int atomic) {
        if (n <= 0) return;
        hipLaunchKernelGGL((k_csdm_mfma<true>), dim3((unsigned)n, nsb, 1), dim3(256), 0, c.stream, Xt, nch, nchp, mp, nsb, G, f,
                           u0, sl, atomic);
        if (nsb > 1)
            hipLaunchKernelGGL((k_csdm_mfma<false>), dim3((unsigned)n, npair, 1), dim3(256), 0, c.stream, Xt, nch, nchp, mp,
                               nsb, G, f, u0, sl, atomic);
    };
    go(0, main_units, slices, fs, slices > 1);
    go(main_units * 8, (units - main_units) * 8, slices * 8, fs / 8, 1);
    return 0;
}

// fused path (nch <= 64: one channel superblock): bins [0, 16*ngroups) straight from Xs; the remaining 1..16 bins
// through a gathered copy + k_csdm_mfma
int launch_csdm_fused(LaunchCtx c, const cf *Xs, cf *Xt_tail, int nch, int64_t m, int nb, double *G, int ld) {
    if (ld <= 0) ld = nb;                                  // row pitch of the spectra (>= nb)
    const int nchp = (nch + 63) / 64 * 64;
    const int ngroups = (nb - 1) / SP_CMF_BINS;
    if (ngroups > 0) {
        // frame slices (multiples of 8 frames) when there are fewer bin groups than CUs; sliced units add atomically
        int slices = (c.ncu + ngroups - 1) / ngroups;
        if (const char *e = getenv("SP_CSDM_SLICES")) slices = atoi(e);           // experiments
        const int max_slices = (int)((m + 127) / 128);
        if (slices > max_slices) slices = max_slices;
        if (slices < 1) slices = 1;
        int64_t fs = (m + slices - 1) / slices;
        fs = (fs + SP_CMF_F - 1) / SP_CMF_F * SP_CMF_F;
        slices = (int)((m + fs - 1) / fs);
        const size_t lds = 2 * sizeof(cf) * SP_CMF_TILE;                          // 68 KiB
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute((const void *)k_csdm_fused, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_done = true;
        }
        hipLaunchKernelGGL(k_csdm_fused, dim3(ngroups * slices), dim3(1024), lds, c.stream, Xs, nch, m, ld, G, fs, slices,
                           slices > 1);
    }
    const int kfirst = SP_CMF_BINS * ngroups, ntail = nb - kfirst;
    if (ntail > 0) {
        const int64_t mp = (m + 31) / 32 * 32;
        hipLaunchKernelGGL(k_csdm_gather_bins, dim3((unsigned)mp), dim3(256), 0, c.stream, Xs, Xt_tail, nch, nchp, m, mp, ld, kfirst,
                           ntail);
        if (launch_csdm_mfma(c, Xt_tail, nch, nchp, mp, ntail, G + (int64_t)kfirst * nch * nch * 2)) return -1;
    }
    return 0;
}

}   // namespace sp
