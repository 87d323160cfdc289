// k_welch.hip -- fused Welch PSD launchers (generic + register-carried metric kernel)
#include "launch.h"
namespace sp {

template <int N, bool CPLX>
static bool try_carry(LaunchCtx c, const void *x, const float *win, int hop, int64_t nframes, const float *trend,
                      const Xf &xf, float *partial, const RunPart &rp) {
    using C = WgCfg<N>;
    if (hop % C::T != 0) return false;
    const int shift = hop / C::T;
#define CARRY_(S)                                                                                     \
    case S:                                                                                           \
        hipLaunchKernelGGL((k_welch_carry<N, CPLX, S>), dim3(rp.blocks), dim3(C::WG), C::lds_bytes(1), c.stream, x, win, \
                           nframes, rp.fpg, trend, xf.tb, partial);                                   \
        return true;
    switch (shift) {
        CARRY_(4) CARRY_(8) CARRY_(16)
        default: return false;
    }
#undef CARRY_
}

int launch_welch(LaunchCtx c, const void *x, bool cplx, const float *win, int hop, int64_t nframes, const float *trend,
                 bool lin, const Xf &xf, float *partial, const RunPart &rp, bool allow_carry, const char **kname) {
    if (kname) *kname = "k_welch";
    if (allow_carry && !lin && !xf.blue) {
        bool done = false;
#define TRY_(NN)                                                                                      \
    case NN:                                                                                          \
        done = cplx ? try_carry<NN, true>(c, x, win, hop, nframes, trend, xf, partial, rp)            \
                    : try_carry<NN, false>(c, x, win, hop, nframes, trend, xf, partial, rp);          \
        break;
        switch (xf.L) {
            TRY_(256) TRY_(512) TRY_(1024) TRY_(2048) TRY_(4096) TRY_(8192)
            default: break;
        }
#undef TRY_
        if (done) {
            if (kname) *kname = "k_welch_carry";
            return 0;
        }
    }
#define M_(XT)                                                                                        \
    if (cplx) {                                                                                       \
        if (lin) hipLaunchKernelGGL((k_welch<XT, true, true>), dim3(rp.blocks), dim3(XT::C::WG), XT::C::lds_bytes(1), \
                                    c.stream, x, win, hop, nframes, rp.fpg, trend, xf.tb, partial);   \
        else hipLaunchKernelGGL((k_welch<XT, true, false>), dim3(rp.blocks), dim3(XT::C::WG), XT::C::lds_bytes(1), \
                                c.stream, x, win, hop, nframes, rp.fpg, trend, xf.tb, partial);       \
    } else {                                                                                          \
        if (lin) hipLaunchKernelGGL((k_welch<XT, false, true>), dim3(rp.blocks), dim3(XT::C::WG), XT::C::lds_bytes(1), \
                                    c.stream, x, win, hop, nframes, rp.fpg, trend, xf.tb, partial);   \
        else hipLaunchKernelGGL((k_welch<XT, false, false>), dim3(rp.blocks), dim3(XT::C::WG), XT::C::lds_bytes(1), \
                                c.stream, x, win, hop, nframes, rp.fpg, trend, xf.tb, partial);       \
    }
    SP_DISPATCH_X(xf, M_)
#undef M_
    return 0;
}

int launch_welch_finish(LaunchCtx c, const float *partial, int64_t G, const Xf &xf, int sided, double scale, double *out) {
    const int n = xf.tb.n;
    hipLaunchKernelGGL(k_welch_finish, dim3((n + SP_FIN_BINS - 1) / SP_FIN_BINS), dim3(SP_FIN_BINS * SP_FIN_SLICES), 0,
                       c.stream, partial, G, xf.L, n, sided, scale, out);
    return 0;
}

}   // namespace sp
