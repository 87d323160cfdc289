// k_stft.hip -- STFT / spectrogram frame launchers
#include "launch.h"
namespace sp {

int launch_stft(LaunchCtx c, const void *x, bool cplx, const float *win, int hop, int64_t nframes, const float *trend,
                bool lin, const Xf &xf, const RunPart &rp, int sided, float amp, int out_power, void *out, double *pseg) {
#define L_(XT, CP, LN)                                                                                \
    hipLaunchKernelGGL((k_stft<XT, CP, LN>), dim3(rp.blocks), dim3(XT::C::WG), XT::C::lds_bytes(1), c.stream, x, win, hop, \
                       nframes, rp.fpg, trend, xf.tb, sided, amp, out_power, out, pseg)
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

}   // namespace sp
