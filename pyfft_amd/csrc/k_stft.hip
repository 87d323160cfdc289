// k_stft.hip -- STFT / spectrogram frame launchers
#include "launch.h"
namespace sp {

int launch_stft(LaunchCtx c, const void *x, bool cplx, const float *win, int hop, int64_t nframes, const float *trend,
                bool lin, const Xf &xf, const RunPart &rp, int sided, float amp, int out_power, void *out, double *pseg,
                int segmean, cf *cog, int klo, int khi) {
#define L_(XT, CP, LN)                                                                                \
    hipLaunchKernelGGL((k_stft<XT, CP, LN>), dim3(rp.blocks), dim3(XT::C::WG), XT::C::lds_bytes(1), c.stream, x, win, hop, \
                       nframes, rp.fpg, trend, xf.tb, sided, amp, out_power, out, pseg, segmean, cog, klo, khi)
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

// real input, power-of-two n >= 32: two frames per transform; rp partitions PAIRS of frames
// groups per CU for k_stft_rp's run partition: twice what the selected instantiation keeps resident when that is three or more
// (the compile-time one-sided form with the window in LDS), else the default 4 (one or two resident: whole rounds)
int stft_rp_groups_per_cu(const Xf &xf, bool lin, int hop, int sided, int out_power, bool pseg) {
    if (xf.blue || lin || !(sided == SIDED_ONE && out_power == 0 && !pseg) || getenv("SP_STFT_NOFAST") || getenv("SP_GROUPS_PER_CU")) return 0;
    const int T_ = xf.L / 16;
    const bool s4 = xf.L >= 1024 && hop % T_ == 0 && hop / T_ == 4 && !getenv("SP_STFT_NOCARRY");
    int res = 0;
#define RQ_(NN)                                                                                       \
    case NN:                                                                                          \
        res = s4 ? resident_per_cu((const void *)k_stft_rp<NN, false, 4, 1>, WgCfg<NN>::WG, WgCfg<NN>::lds_bytes(1) + (SP_STFT_WLDS ? sizeof(float) * NN : 0)) \
                 : resident_per_cu((const void *)k_stft_rp<NN, false, 0, 1>, WgCfg<NN>::WG, WgCfg<NN>::lds_bytes(1) + (SP_STFT_WLDS ? sizeof(float) * NN : 0)); \
        break;
    switch (xf.L) {
        RQ_(1024) RQ_(2048) RQ_(4096) RQ_(8192)
        default: return 0;
    }
#undef RQ_
    return res >= 3 ? 2 * res : 0;
}

int launch_stft_rp(LaunchCtx c, const float *x, const float *win, int hop, int64_t nframes, const float *trend, bool lin,
                   const Xf &xf, const RunPart &rp, int sided, float amp, int out_power, void *out, double *pseg, int nchan,
                   int64_t x_cs, int64_t out_cs, int out_ld) {
    // register-carried overlap (k_stft_rp<.., SHIFT>) for the long transforms at 75 % overlap without linear detrend: -5..-10 %
    // (tools/stftcarry_ab.py; at 50 % overlap the same form measured 40 % SLOWER than the plain double fetch and is not used)
    const int T_ = xf.L / 16;
    const int shift = (!lin && xf.L >= 1024 && hop % T_ == 0 && hop / T_ == 4 && !getenv("SP_STFT_NOCARRY")) ? 4 : 0;
#define RPS_(NN, S)                                                                                   \
    hipLaunchKernelGGL((k_stft_rp<NN, false, S>), dim3(rp.blocks, nchan), dim3(WgCfg<NN>::WG), WgCfg<NN>::lds_bytes(1), c.stream, x, \
                       win, hop, nframes, rp.fpg, trend, xf.tb, sided, amp, out_power, out, pseg, x_cs, out_cs, out_ld)
    // the one-sided complex spectrogram without per-frame power: compile-time form (k_stft_rp<.., FAST>)
    const bool fast = !lin && sided == SIDED_ONE && out_power == 0 && pseg == nullptr && !getenv("SP_STFT_NOFAST");
#define RPF_(NN, S)                                                                                   \
    hipLaunchKernelGGL((k_stft_rp<NN, false, S, 1>), dim3(rp.blocks, nchan), dim3(WgCfg<NN>::WG),     \
                       WgCfg<NN>::lds_bytes(1) + (SP_STFT_WLDS ? sizeof(float) * NN : 0), c.stream, x,    \
                       win, hop, nframes, rp.fpg, trend, xf.tb, sided, amp, out_power, out, pseg, x_cs, out_cs, out_ld)
#define RPC_(NN)                                                                                      \
    case NN:                                                                                          \
        if (fast && shift == 4) RPF_(NN, 4);                                                          \
        else if (fast) RPF_(NN, 0);                                                                   \
        else if (shift == 4) RPS_(NN, 4);                                                             \
        else if (lin) hipLaunchKernelGGL((k_stft_rp<NN, true>), dim3(rp.blocks, nchan), dim3(WgCfg<NN>::WG),     \
                                         WgCfg<NN>::lds_bytes(1), c.stream, x, win, hop, nframes, rp.fpg, trend, xf.tb, sided, \
                                         amp, out_power, out, pseg, x_cs, out_cs, out_ld);            \
        else RPS_(NN, 0);                                                                             \
        break;
#define RP_(NN)                                                                                       \
    case NN:                                                                                          \
        if (lin) hipLaunchKernelGGL((k_stft_rp<NN, true>), dim3(rp.blocks, nchan), dim3(WgCfg<NN>::WG),          \
                                    WgCfg<NN>::lds_bytes(1), c.stream, x, win, hop, nframes, rp.fpg, trend, xf.tb, sided, \
                                    amp, out_power, out, pseg, x_cs, out_cs, out_ld);                 \
        else hipLaunchKernelGGL((k_stft_rp<NN, false>), dim3(rp.blocks, nchan), dim3(WgCfg<NN>::WG),             \
                                WgCfg<NN>::lds_bytes(1), c.stream, x, win, hop, nframes, rp.fpg, trend, xf.tb, sided, amp, \
                                out_power, out, pseg, x_cs, out_cs, out_ld);                          \
        break;
    switch (xf.L) {
        RP_(32) RP_(64) RP_(128) RP_(256) RP_(512) RPC_(1024) RPC_(2048) RPC_(4096) RPC_(8192)
        default: return -1;
    }
#undef RP_
#undef RPC_
#undef RPS_
#undef RPF_
    return 0;
}

int launch_cog_finish_op(LaunchCtx c, const cf *acc, int wpf, int64_t nframes, double df, double *out, const cf *lobe, CogLobe lb,
                         const double *st, const float *trend, int64_t nmean, int n) {
    hipLaunchKernelGGL(k_cog_finish_op, dim3((unsigned)((nframes + 255) / 256)), dim3(256), 0, c.stream, acc, wpf, nframes, df, out,
                       lobe, lb, st, trend, nmean, n);
    return 0;
}
int launch_cog_finish(LaunchCtx c, const cf *acc, int wpf, int64_t nframes, double df, double *out) {
    hipLaunchKernelGGL(k_cog_finish, dim3((unsigned)((nframes + 255) / 256)), dim3(256), 0, c.stream, acc, wpf, nframes, df, out);
    return 0;
}

}   // namespace sp
