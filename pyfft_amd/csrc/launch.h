// launch.h -- host-side launchers, one translation unit per kernel family so they compile in parallel.
#pragma once
#include <stdlib.h>
#include "kernels.h"
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

namespace sp {

struct LaunchCtx {
    hipStream_t stream;
    int ncu;
    // optional: recorded as the launch's own completion signal (hipExtLaunchKernelGGL) by the launchers that honour it
    // (launch_welch_pipe) -- a hipEventRecord behind the kernel is a packet of its own and costs the next launch on the
    // stream 3-5 us (tools/ubench/coexec_rccl.hip)
    hipEvent_t stop = nullptr;
};

// transform selection for a length n: pow2 workgroup FFT, or Bluestein on an L-point one
struct Xf {
    XfTables tb;
    int L;
    bool blue;
};

inline int fpw_of(int L) {
    const int R = L < 16 ? L : 16, T = L / R, WG = T >= 256 ? T : 256;
    return WG / T;
}

// frames are dealt to transform groups in contiguous runs
struct RunPart {
    int64_t groups, fpg;
    int blocks;
};
// groups_per_cu: 4 by default = two rounds at the 2 resident workgroups per CU of the 200-VGPR segment kernels.  Measured
// on the metric shape (bench.py step / kernel, ms): 2 -> 0.651 / 0.607, 4 -> 0.660 / 0.605, 8 -> 0.680 / 0.634,
// 3 -> 0.73 / 0.685 (one and a half rounds), 6 ~ 8, 12 and 24 worse: longer runs amortise the per-workgroup prologue
// (twiddle constants, first frame) and halve the partial spectra the epilogue has to sum.  SP_GROUPS_PER_CU overrides it.
inline int default_groups_per_cu() {
    static const int v = [] {
        const char *e = getenv("SP_GROUPS_PER_CU");
        const int k = e ? atoi(e) : 0;
        return k > 0 ? k : 4;
    }();
    return v;
}
inline RunPart run_partition(int L, int64_t nframes, int ncu, int groups_per_cu = 0) {
    if (groups_per_cu <= 0) groups_per_cu = default_groups_per_cu();
    const int fpw = fpw_of(L);
    const int64_t target = (int64_t)ncu * groups_per_cu * fpw;
    int64_t f = (nframes + target - 1) / target;
    if (f < 1) f = 1;
    if (const char *e = getenv("SP_FPG1")) {              // experiments: frames per group of the one-dimensional partitions
        const int64_t v = atoll(e);
        if (v > 0) f = v;
    }
    const int64_t G = (nframes + f - 1) / f;
    RunPart r;
    r.fpg = f;
    r.blocks = (int)((G + fpw - 1) / fpw);
    r.groups = (int64_t)r.blocks * fpw;
    return r;
}
// the same for kernels whose grid has a second dimension of `ny` independent rows (channels): about 8 workgroups per
// CU in total, so that the per-group partial spectra (ny x groups x 3..4 x L floats) stay small -- with 63 channels the
// per-CU rule above wrote and re-read 1.6 GB of partials
// slots (optional) = workgroups the whole GPU keeps resident for this kernel: the group count per row is then rounded so that the
// grid is just under a whole number of rounds (63 channels x 33 groups = 2079 workgroups over 512 slots ran a fifth round of 31;
// 32 groups = 2016 run four: 2.51 -> 2.41 ms for the reference against 63 channels of 2^24 samples)
inline RunPart run_partition_2d(int L, int64_t nframes, int ncu, int ny, int64_t slots = 0) {
    const int fpw = fpw_of(L);
    int64_t target = ((int64_t)ncu * 8 + ny - 1) / (ny > 0 ? ny : 1) * fpw;
    if (target < 8 * fpw) target = 8 * fpw;
    int64_t f = (nframes + target - 1) / target;
    if (f < 1) f = 1;
    if (slots > 0 && ny > 0) {
        const int64_t wg = ((nframes + f - 1) / f + fpw - 1) / fpw * ny;          // workgroups of the default rule
        int64_t rounds = (wg + slots / 2) / slots;
        if (rounds < 1) rounds = 1;
        const int64_t groups = rounds * slots / ny * fpw;                          // groups per row that fill `rounds` rounds
        if (groups >= 1) {
            const int64_t f2 = (nframes + groups - 1) / groups;
            if (f2 >= 1 && f2 <= 4 * f) f = f2;
        }
    }
    if (const char *e = getenv("SP_FPG")) {               // experiments: frames per group
        const int64_t v = atoll(e);
        if (v > 0) f = v;
    }
    const int64_t G = (nframes + f - 1) / f;
    RunPart r;
    r.fpg = f;
    r.blocks = (int)((G + fpw - 1) / fpw);
    r.groups = (int64_t)r.blocks * fpw;
    return r;
}
// grid of the row kernels (FFT rows, Hilbert rows, FIR block pairs): every workgroup pays a prologue (twiddle constants with
// 30 divisions per twiddled pass) before its first row, so few, long-lived workgroups win for the long transforms -- measured
// (tools/capsweep.py, ms): 4096 rows of 4096: 0.088 at 16 blocks per CU, 0.062 at 4, 0.055 at 2; 65536 rows: 0.825 / 0.790 /
// 0.823; 8192 points: 0.245 / 0.170 / 0.160; 2048 points: 4 per CU wins for 4096 rows (0.039 vs 0.050), 16 for 65536 rows
// (0.394 vs 0.413); <= 1024 points: no difference.  Rule: 4 per CU from 4096 points up; at 2048 points at least 4 rows per
// workgroup (not below 2 per CU); 16 per CU otherwise.  SP_STRIDED_CAP overrides it.
// per_cu: the cap for L >= 4096 in workgroups per CU -- a multiple of what the kernel keeps resident per CU, so that the last
// round of workgroups is a full one (round 3: k_fft_c2c<4096> holds 3 per CU; 4 per CU = 1024 workgroups ran 768 + a 256 tail:
// 0.88 against 0.79 ms at 12 per CU for 65536 rows; the batched Hilbert, 4 rows per workgroup at 4096 rows: 0.065 -> 0.060 at 3)
inline int strided_blocks(int L, int64_t items, int ncu, int per_cu = 4) {
    static const int forced = [] {
        const char *e = getenv("SP_STRIDED_CAP");
        return e ? atoi(e) : 0;
    }();
    const int fpw = fpw_of(L);
    int64_t b = (items + fpw - 1) / fpw;
    int64_t cap = (int64_t)ncu * (forced > 0 ? forced : (L >= 4096 ? per_cu : 16));
    if (forced <= 0 && L == 2048) {
        const int64_t q = b / 4;
        if (q < cap) cap = q > (int64_t)ncu * 2 ? q : (int64_t)ncu * 2;
    }
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// workgroups of `fn` one CU keeps resident (registers, LDS, wave slots; asked once per kernel from the runtime): grids and run
// partitions are sized as a multiple of it, so that the last round of workgroups is a full one
int resident_per_cu(const void *fn, int threads, size_t lds_bytes);
// groups per CU for k_welch_rp's run partition at this transform (a multiple of what the selected instantiation keeps resident)
int welch_rp_groups_per_cu(const Xf &xf, bool lin);
int csd_pair_resident(const Xf &xf, bool lin, bool onepass);
int stft_rp_groups_per_cu(const Xf &xf, bool lin, int hop, int sided, int out_power, bool pseg);

// every launcher returns 0 or -1 (unsupported L); kernel launch errors surface through hipGetLastError
int launch_fft_c2c(LaunchCtx c, const cf *in, cf *out, int64_t batch, int inverse, const Xf &xf,
                   BigTw bt = BigTw{nullptr, nullptr, 0, 0});
int launch_fft_strided(LaunchCtx c, const cf *in, cf *out, int64_t batch, int64_t in_rs, int64_t in_es, int64_t out_rs,
                       int64_t out_es, int conj_in, int conj_out, float scale, const Xf &xf, BigTw bt);
int launch_fft_cols(LaunchCtx c, const cf *in, cf *out, int64_t ncols, int64_t nouter, int64_t es, int64_t os, int64_t twmul,
                    int conj_in, const Xf &xf, BigTw bt, int64_t hmask_n = 0,
                    ColsIn ci = ColsIn{0, nullptr, nullptr, nullptr, 0}, int tw_outer = 0);
int launch_fft_rows_rev(LaunchCtx c, const cf *in, cf *out, int64_t A, int64_t B, int conj_out, float scale, const Xf &xf,
                        RowsOut ro = RowsOut{nullptr, 0, 0, nullptr});
// elementwise / transpose pieces of the long paths (k_fft.hip)
int launch_transpose_c(LaunchCtx c, const cf *in, cf *out, int64_t rows, int64_t cols, int conj, float scale, int64_t batch = 1);
int launch_pack_real(LaunchCtx c, const float *x, int64_t n_in, const double *mean, int64_t L, cf *out);
int launch_cmul_vec(LaunchCtx c, const cf *a, const cf *b, int64_t n, int conj_out, cf *out, int64_t batch = 1);
int launch_blue_pre(LaunchCtx c, const cf *in, const cf *chirp, int64_t n, int64_t L, int conj_in, cf *out, int64_t batch = 1);
int launch_blue_post(LaunchCtx c, const cf *in, const cf *chirp, int64_t n, int conj_out, float scale, cf *out, int64_t batch = 1,
                     int64_t in_ld = 0);
int launch_hilbert_mask(LaunchCtx c, cf *X, int64_t n);
int launch_xc_pack(LaunchCtx c, const float *x1, const float *x2, int64_t n, int64_t L, const double *mom, cf *z);
int launch_xc_mid(LaunchCtx c, const cf *Z, int64_t L, cf *R);
int launch_xc_mid_half(LaunchCtx c, const cf *Z, int64_t L, BigTw bt, cf *Zp);
int launch_xc_out(LaunchCtx c, const cf *r, int64_t n, int64_t L, const double *mom, float *co);
bool welch_carry_eligible(const Xf &xf, int hop, bool lin);
int launch_welch(LaunchCtx c, const void *x, bool cplx, const float *win, int hop, int64_t nframes, const float *trend,
                 bool lin, const Xf &xf, float *partial, const RunPart &rp, bool allow_carry, cf *spartial,
                 const char **kname, int segmean = 0);
bool welch_pipe_eligible(const Xf &xf, int hop);
int launch_welch_pipe(LaunchCtx c, const void *x, bool cplx, const float *win, int hop, int64_t nframes, float *trend,
                      const Xf &xf, float *partial, const RunPart &rp, cf *spartial, int mode = 0, int nch = 1, int64_t x_cs = 0,
                      int gpr = 0);
int launch_op_estimate(LaunchCtx c, const void *x, bool cplx, int64_t nsig, double *part, float *trend);
int launch_op_reduce(LaunchCtx c, const void *x, bool cplx, const float *trend, const float *partial, const cf *spartial,
                     int64_t G, const Xf &xf, int hop, int64_t nframes, int64_t nmean, OnePass st, double *sum_out);
int launch_op_finish(LaunchCtx c, const void *x, bool cplx, const float *trend, const float *win, OnePass st,
                     const double *mean_in, int64_t nmean, const Xf &xf, int hop, int64_t nframes, cf *cw, const cf *Wf,
                     int sided, double scale, double *out, bool export_state = false);
int launch_op_apply(LaunchCtx c, const double *state, const cf *Wf, int n, int sided, double scale, double *out);
// colsums + finish (or export) in one launch for a window whose spectrum is confined to the bins -3 .. 3 (k_op_fused)
int launch_op_fused(LaunchCtx c, const void *x, bool cplx, const float *trend, const float *win, const float *partial,
                    const cf *spartial, int64_t G, int n, int hop, int64_t nframes, int64_t nmean, OnePass st, unsigned *ticket,
                    CogLobe lb, const double *mean_in, int sided, double scale, double *out, bool export_state,
                    OpPrev prev = OpPrev{nullptr, nullptr, nullptr, 0, 0.0}, bool light = false, double cola_c = 0.0);
int launch_welch_finish(LaunchCtx c, const float *partial, int64_t G, const Xf &xf, int sided, double scale, double *out,
                        int sym = 0);
int launch_welch_rp(LaunchCtx c, const float *x, const float *win, int hop, int64_t nframes, const float *trend, bool lin,
                    const Xf &xf, float *partial, const RunPart &rp);
int launch_stft_rp(LaunchCtx c, const float *x, const float *win, int hop, int64_t nframes, const float *trend, bool lin,
                   const Xf &xf, const RunPart &rp, int sided, float amp, int out_power, void *out, double *pseg,
                   int nchan = 1, int64_t x_cs = 0, int64_t out_cs = 0, int out_ld = 0);
int launch_csd(LaunchCtx c, const void *x, const void *y, bool cplx, int nch, int64_t y_ld, const float *win, int hop,
               int64_t nframes, const float *trend_x, const float *trend_y, bool lin, const Xf &xf, float *partial,
               const RunPart &rp, int segmean = 0);
int launch_csd_finish(LaunchCtx c, const float *partial, int64_t G, const Xf &xf, int nch, int sided, double scale,
                      double *pxx, double *pyy, double *pxy);
int launch_stft(LaunchCtx c, const void *x, bool cplx, const float *win, int hop, int64_t nframes, const float *trend,
                bool lin, const Xf &xf, const RunPart &rp, int sided, float amp, int out_power, void *out, double *pseg, int segmean = 0,
                cf *cog = nullptr, int klo = 0, int khi = 0);
int launch_cog_finish(LaunchCtx c, const cf *acc, int wpf, int64_t nframes, double df, double *out);
int launch_cog_carry(LaunchCtx c, const void *x, bool cplx, const float *win, int hop, int64_t nframes, const float *trend,
                     const Xf &xf, cf *cog, const RunPart &rp, int klo, int khi);
bool csd_rp_eligible(const Xf &xf);
int launch_csd_rp(LaunchCtx c, const float *x, const float *y, int nch, int64_t y_ld, const float *win, int hop,
                  int64_t nframes, const float *trend_x, const float *trend_y, bool lin, const Xf &xf, float *partial,
                  const RunPart &rp);
int launch_csd_rp_finish(LaunchCtx c, const float *partial, int64_t G, const Xf &xf, int nch, int sided, double scale,
                         double *pxx, double *pyy, double *pxy);
int launch_pairspec(LaunchCtx c, const float *x, const float *win, int hop, int64_t nframes, const float *trend, bool lin,
                    const Xf &xf, const RunPart &rp, cf *Zx);
int launch_csd_pair(LaunchCtx c, const float *y, int nch, int64_t y_ld, const float *win, int hop, int64_t nframes,
                    float *trend_y, bool lin, const Xf &xf, const cf *Zx, float *partial, const RunPart &rp, cf *spartial = nullptr);
int launch_csd_pair_finish(LaunchCtx c, const float *partial, int64_t G, const Xf &xf, int nch, int sided, double scale,
                           double *pyy, double *pxy, const double *st_y = nullptr, const double *st_x = nullptr, const cf *Wf = nullptr,
                           const float *trend_x = nullptr, const float *trend_y = nullptr, int64_t nmean = 0, int64_t M = 0);
#define SP_COLSUM_SLICES 256
int launch_colsum_real(LaunchCtx c, const float *x, const float *trend, int H, int64_t M, cf *out);
int launch_csdm_transpose(LaunchCtx c, const cf *Xs, cf *Xt, int nch, int64_t mc, int nb);
int launch_csdm_gemm(LaunchCtx c, const cf *Xt, int nch, int64_t mc, int nb, double *G);
int launch_csdm_finish(LaunchCtx c, double *G, int nch, int nb, double scale, int blk);
int launch_csdm_transpose_kgc(LaunchCtx c, const cf *Xs, cf *Xt, int nch, int nchp, int64_t m, int64_t mp, int nb);
int launch_csdm_mfma(LaunchCtx c, const cf *Xt, int nch, int nchp, int64_t mp, int nb, double *G);
int launch_csdm_bf16(LaunchCtx c, const cf *Xs, cf *Xt_tail, int nch, int64_t m, int nb, double *G, int ld, int two_pieces = 0, int init = 0);
int launch_csdm_fold(LaunchCtx c, const double *H, double *G, int nch, int n, const double *st = nullptr, const cf *Wf = nullptr,
                     const float *trend = nullptr, int64_t nmean = 0, int64_t M = 0, double scale = 1.0, int init = 0);
int launch_cm_blocksums(LaunchCtx c, const cf *spartial, int nch, int runs, int hop, double *Sl);
int launch_op_finish_channels(LaunchCtx c, const void *x, int64_t x_cs, int nch, const float *trend, const float *win,
                              const double *Sl, const cf *Wf, int hop, int64_t nframes, int64_t nmean, const Xf &xf, double *out,
                              bool cplx = false);
int launch_cog_finish_op(LaunchCtx c, const cf *acc, int wpf, int64_t nframes, double df, double *out, const cf *lobe, CogLobe lb,
                         const double *st, const float *trend, int64_t nmean, int n);
int launch_csdm_fused(LaunchCtx c, const cf *Xs, cf *Xt_tail, int nch, int64_t m, int nb, double *G, int ld = 0);
int launch_hilbert_mid(LaunchCtx c, cf *Z, int64_t M, BigTw bt);
// half-length Hilbert with the middle step inside the row pass + the two adjoint column passes (k_hilbert_rowsmid, k_fft_cols_inv)
int launch_hilbert_rowsmid(LaunchCtx c, cf *Tm, int64_t A, int64_t B, const Xf &xc, BigTw btN, const cf *tw2c);
// long ccf with the middle step and the half-length transform's first pass inside the row pass (k_xc_rowsmid, k_fft_cols_lag)
int launch_xc_rowsmid(LaunchCtx c, cf *Tm, int64_t A, int64_t B, const Xf &xc, const Xf &xc2, BigTw btL, BigTw btM);
int launch_fft_cols_lag(LaunchCtx c, const cf *in, int64_t ncols, int64_t nouter, int64_t es, int64_t os, const Xf &xf, RowsOut ro);
int launch_fft_cols_inv(LaunchCtx c, const cf *in, cf *out, int64_t ncols, int64_t nouter, int64_t es, int64_t os, int64_t twmul,
                        const Xf &xf, BigTw bt, float scale, const RowsOut *analytic);
int launch_hilbert(LaunchCtx c, const float *x, int64_t n_in, int64_t x_ld, int64_t batch, const Xf &xf, cf *out,
                   const cf *H = nullptr);
int launch_spec_mul(LaunchCtx c, cf *X, const cf *H, int64_t n);
int frame_sum_slices(int ncu, int nch, int nfft, int64_t nframes);
int launch_frame_sum(LaunchCtx c, const void *x, bool cplx, int64_t x_ld, int nch, int nfft, int hop, int64_t nframes,
                     const float *trend, bool lin, double *out, double *part);
int launch_fftfilt(LaunchCtx c, const float *x, int64_t n, int ntaps, const cf *Hs, const Xf &xf, float *y);
int launch_xcorr(LaunchCtx c, const float *x1, const float *x2, int64_t n, const double *mom, const Xf &xf, float *co);
int launch_moments(LaunchCtx c, const void *x, bool cplx, int64_t n, int mode, double *partial_scratch, double *out_d,
                   float *trend_f, int nsignals = 1, int64_t x_cs = 0);
int launch_moments_xc(LaunchCtx c, const float *x, int64_t n, double *partial, double *out_d, double *xc_out, int64_t x_cs);
int launch_transpose(LaunchCtx c, const void *in, void *out, int64_t rows, int64_t cols, int elem_bytes);
// segments longer than one workgroup transform (k_long.hip); m frames starting at frame f0, rows S[m][nfft]
int launch_long_segstats(LaunchCtx c, const void *x, bool cplx, int64_t f0, int64_t m, int hop, int nfft, int mode, float *rec);
int launch_long_pack(LaunchCtx c, const void *x, bool cplx, const float *win, int nfft, int hop, int64_t f0, int64_t m,
                     const float *trend, bool lin, const float *segrec, cf *S, double *pseg);
int launch_long_acc_psd(LaunchCtx c, const cf *S, int64_t m, int nfft, double *acc);
int launch_long_acc_csd(LaunchCtx c, const cf *Sx, const cf *Sy, int64_t m, int nfft, double *ayy, double *axy);
int launch_long_finish(LaunchCtx c, const double *acc, int nfft, int sided, double scale, bool cplx, double *out);
int launch_long_stft_out(LaunchCtx c, const cf *S, int64_t m, int nfft, int sided, float amp, int out_power, void *out,
                         int64_t f0, int nb);
int launch_long_cog(LaunchCtx c, const cf *S, int64_t m, int nfft, int klo, int khi, cf *acc, int64_t f0);

// exact second-order IIR section (k_iir.hip)
int64_t biquad_tiles(int64_t n);
int launch_biquad(LaunchCtx c, const double *b, const double *a, const float *x, int64_t n, float *y, double *work);

// fft_pwelch epilogue on device-resident spectra (k_epilogue.hip)
int launch_epi_elem(LaunchCtx c, const double *pxx, const double *pyy, const double *pxy, int nch, int nb, int nfft, int onesided,
                    double enbw, double *cxy, double *cxy2, double *phi, double *lxx, double *lyy, double *lxy);
int launch_epi_spec(LaunchCtx c, const double *pxx, const double *pyy, const double *pxy, const double *cxy, int nch, int nb,
                    int nfft, int onesided, cf *X, double *rowmax /* [1 + nch] scratch: the rows' scales */);
int launch_epi_corr(LaunchCtx c, const cf *X, int nch, int nfft, int onesided, double *rxx, double *ryy, double *rxy, double *icxy,
                    double *ee, double *cc, const double *rowmax);

// dispatch over the transform: MACRO(XTYPE) with XTYPE = XfPow2<L> or XfBlue<L>
#define SP_CASE_P(Lv, MACRO) case Lv: { MACRO(XfPow2<Lv>) } break;
#define SP_CASE_B(Lv, MACRO) case Lv: { MACRO(XfBlue<Lv>) } break;
#define SP_DISPATCH_X(xf, MACRO)                                                                      \
    if (!(xf).blue) {                                                                                 \
        switch ((xf).L) {                                                                             \
            SP_CASE_P(2, MACRO) SP_CASE_P(4, MACRO) SP_CASE_P(8, MACRO) SP_CASE_P(16, MACRO)          \
            SP_CASE_P(32, MACRO) SP_CASE_P(64, MACRO) SP_CASE_P(128, MACRO) SP_CASE_P(256, MACRO)     \
            SP_CASE_P(512, MACRO) SP_CASE_P(1024, MACRO) SP_CASE_P(2048, MACRO) SP_CASE_P(4096, MACRO) \
            SP_CASE_P(8192, MACRO)                                                                    \
            default: return -1;                                                                       \
        }                                                                                             \
    } else {                                                                                          \
        switch ((xf).L) {                                                                             \
            SP_CASE_B(16, MACRO) SP_CASE_B(32, MACRO) SP_CASE_B(64, MACRO) SP_CASE_B(128, MACRO)      \
            SP_CASE_B(256, MACRO) SP_CASE_B(512, MACRO) SP_CASE_B(1024, MACRO) SP_CASE_B(2048, MACRO) \
            SP_CASE_B(4096, MACRO) SP_CASE_B(8192, MACRO)                                             \
            default: return -1;                                                                       \
        }                                                                                             \
    }
#define SP_DISPATCH_P(xf, MACRO)                                                                      \
    switch ((xf).L) {                                                                                 \
        SP_CASE_P(2, MACRO) SP_CASE_P(4, MACRO) SP_CASE_P(8, MACRO) SP_CASE_P(16, MACRO)              \
        SP_CASE_P(32, MACRO) SP_CASE_P(64, MACRO) SP_CASE_P(128, MACRO) SP_CASE_P(256, MACRO)         \
        SP_CASE_P(512, MACRO) SP_CASE_P(1024, MACRO) SP_CASE_P(2048, MACRO) SP_CASE_P(4096, MACRO)    \
        SP_CASE_P(8192, MACRO)                                                                        \
        default: return -1;                                                                           \
    }

}   // namespace sp
