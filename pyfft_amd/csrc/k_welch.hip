// k_welch.hip -- fused Welch PSD launchers (generic + register-carried metric kernel + one-pass detrend epilogue)
#include "launch.h"
#include <map>
#include <mutex>
namespace sp {

bool welch_carry_eligible(const Xf &xf, int hop, bool lin) {
    if (xf.blue || lin || xf.L < 256 || xf.L > 8192) return false;
    const int T = xf.L / 16;
    if (hop % T != 0) return false;
    const int shift = hop / T;
    return shift == 4 || shift == 8 || shift == 16;
}

// SP_CARRY_LDS_PAD=<bytes>: extra dynamic LDS per workgroup, to pin the number of resident workgroups per CU in
// occupancy experiments
static size_t carry_lds_pad() {
    static const size_t v = [] {
        const char *e = getenv("SP_CARRY_LDS_PAD");
        return e ? (size_t)atol(e) : (size_t)0;
    }();
    return v;
}

template <int N, bool CPLX>
static bool try_carry(LaunchCtx c, const void *x, const float *win, int hop, int64_t nframes, const float *trend,
                      const Xf &xf, float *partial, const RunPart &rp, cf *spartial) {
    using C = WgCfg<N>;
    const int shift = hop / C::T;
    constexpr bool HINT = N >= 512;       // see k_welch_carry / k_welch_carry_nh
    size_t lds = C::lds_bytes(SP_CARRY_NBUF) + carry_lds_pad();
#if SP_CARRY_W3
    if (spartial && N == 4096 && C::FPW == 1) lds += sizeof(float) * (2 * (size_t)shift * C::T + 16 * SP_TW1_PITCH);
#endif
#define CARRY_(S)                                                                                     \
    case S:                                                                                           \
        if constexpr (HINT) {                                                                         \
            if (spartial) hipLaunchKernelGGL((k_welch_carry<N, CPLX, S, true>), dim3(rp.blocks), dim3(C::WG), lds, c.stream, \
                                             x, win, nframes, rp.fpg, trend, xf.tb, partial, spartial); \
            else hipLaunchKernelGGL((k_welch_carry<N, CPLX, S, false>), dim3(rp.blocks), dim3(C::WG), lds, c.stream, \
                                    x, win, nframes, rp.fpg, trend, xf.tb, partial, spartial);        \
        } else {                                                                                      \
            if (spartial) hipLaunchKernelGGL((k_welch_carry_nh<N, CPLX, S, true>), dim3(rp.blocks), dim3(C::WG), lds, c.stream, \
                                             x, win, nframes, rp.fpg, trend, xf.tb, partial, spartial); \
            else hipLaunchKernelGGL((k_welch_carry_nh<N, CPLX, S, false>), dim3(rp.blocks), dim3(C::WG), lds, c.stream, \
                                    x, win, nframes, rp.fpg, trend, xf.tb, partial, spartial);        \
        }                                                                                             \
        return true;
    switch (shift) {
        CARRY_(4) CARRY_(8) CARRY_(16)
        default: return false;
    }
#undef CARRY_
}

// centre-of-gravity moments per frame on the carry kernel (power-of-two nfft, hop = nfft/4, /2 or nfft; constant detrend;
// every bin)
template <int N, bool CPLX>
static bool try_carry_cog(LaunchCtx c, const void *x, const float *win, int hop, int64_t nframes, const float *trend,
                          const Xf &xf, cf *cog, const RunPart &rp) {
    using C = WgCfg<N>;
    const int shift = hop / C::T;
#define COG_(S)                                                                                       \
    case S:                                                                                           \
        hipLaunchKernelGGL((k_welch_carry_cog<N, CPLX, S>), dim3(rp.blocks), dim3(C::WG), C::lds_bytes(SP_CARRY_NBUF), \
                           c.stream, x, win, nframes, rp.fpg, trend, xf.tb, cog);                    \
        return true;
    switch (shift) {
        COG_(4) COG_(8) COG_(16)
        default: return false;
    }
#undef COG_
}

// returns 1 when the shape is not eligible (the caller falls back to the generic frame kernel)
int launch_cog_carry(LaunchCtx c, const void *x, bool cplx, const float *win, int hop, int64_t nframes, const float *trend,
                     const Xf &xf, cf *cog, const RunPart &rp, int klo, int khi) {
    if (!welch_carry_eligible(xf, hop, false)) return 1;
    if (klo > 0 || khi < xf.L / 2) return 1;          // the streaming kernel weighs every bin; a band goes the generic way
    bool done = false;
#define TRY_(NN)                                                                                      \
    case NN:                                                                                          \
        done = cplx ? try_carry_cog<NN, true>(c, x, win, hop, nframes, trend, xf, cog, rp)  \
                    : try_carry_cog<NN, false>(c, x, win, hop, nframes, trend, xf, cog, rp); \
        break;
    switch (xf.L) {
        TRY_(256) TRY_(512) TRY_(1024) TRY_(2048) TRY_(4096) TRY_(8192)
        default: break;
    }
#undef TRY_
    return done ? 0 : 1;
}

int launch_welch(LaunchCtx c, const void *x, bool cplx, const float *win, int hop, int64_t nframes, const float *trend,
                 bool lin, const Xf &xf, float *partial, const RunPart &rp, bool allow_carry, cf *spartial,
                 const char **kname, int segmean) {
    if (kname) *kname = "k_welch";
    if (segmean) allow_carry = false;        // per-segment detrend exists only in the generic kernel
    if (allow_carry && welch_carry_eligible(xf, hop, lin)) {
        bool done = false;
#define TRY_(NN)                                                                                      \
    case NN:                                                                                          \
        done = cplx ? try_carry<NN, true>(c, x, win, hop, nframes, trend, xf, partial, rp, spartial)  \
                    : try_carry<NN, false>(c, x, win, hop, nframes, trend, xf, partial, rp, spartial); \
        break;
        switch (xf.L) {
            TRY_(256) TRY_(512) TRY_(1024) TRY_(2048) TRY_(4096) TRY_(8192)
            default: break;
        }
#undef TRY_
        if (done) {
            if (kname) *kname = spartial ? "k_welch_carry(onepass)" : "k_welch_carry";
            return 0;
        }
    }
    if (spartial) return -1;      // one-pass accumulation exists only in the carry kernel
#define M_(XT)                                                                                        \
    if (cplx) {                                                                                       \
        if (lin) hipLaunchKernelGGL((k_welch<XT, true, true>), dim3(rp.blocks), dim3(XT::C::WG), XT::C::lds_bytes(1), \
                                    c.stream, x, win, hop, nframes, rp.fpg, trend, xf.tb, partial, segmean);   \
        else hipLaunchKernelGGL((k_welch<XT, true, false>), dim3(rp.blocks), dim3(XT::C::WG), XT::C::lds_bytes(1), \
                                c.stream, x, win, hop, nframes, rp.fpg, trend, xf.tb, partial, segmean);       \
    } else {                                                                                          \
        if (lin) hipLaunchKernelGGL((k_welch<XT, false, true>), dim3(rp.blocks), dim3(XT::C::WG), XT::C::lds_bytes(1), \
                                    c.stream, x, win, hop, nframes, rp.fpg, trend, xf.tb, partial, segmean);   \
        else hipLaunchKernelGGL((k_welch<XT, false, false>), dim3(rp.blocks), dim3(XT::C::WG), XT::C::lds_bytes(1), \
                                c.stream, x, win, hop, nframes, rp.fpg, trend, xf.tb, partial, segmean);       \
    }
    SP_DISPATCH_X(xf, M_)
#undef M_
    return 0;
}

int launch_welch_finish(LaunchCtx c, const float *partial, int64_t G, const Xf &xf, int sided, double scale, double *out,
                        int sym) {
    const int n = xf.tb.n;
    hipLaunchKernelGGL(k_welch_finish, dim3((n + SP_FIN_BINS - 1) / SP_FIN_BINS), dim3(SP_FIN_BINS * SP_FIN_SLICES), 0,
                       c.stream, partial, G, xf.L, n, sided, scale, out, sym);
    return 0;
}

// smallest power-of-two transform that uses the hinted entry point of k_welch_rp (experiments: -DSP_RP_HINT_MIN=...).
// Round 3: 4096 (was 2048).  The round-1 finding "the 2-wave hint is 13 % faster at 2048 points" was an artefact of the run
// partition: 4 groups per CU against the 3 workgroups the unhinted kernel keeps resident = a last round a third full.  With the
// partition a multiple of the residency (welch_rp_groups_per_cu) the plain form wins: 0.281 -> 0.252 ms at 2^26 samples, nfft 2048,
// 75 % overlap (tools/rp_ab.sh).
#ifndef SP_RP_HINT_MIN
#define SP_RP_HINT_MIN 4096
#endif
int resident_per_cu(const void *fn, int threads, size_t lds_bytes) {
    struct Key {
        const void *fn;
        int threads;
        size_t lds;
        bool operator<(const Key &o) const { return fn != o.fn ? fn < o.fn : (threads != o.threads ? threads < o.threads : lds < o.lds); }
    };
    static std::mutex mu;
    static std::map<Key, int> cache;
    std::lock_guard<std::mutex> lk(mu);
    const Key k{fn, threads, lds_bytes};
    auto it = cache.find(k);
    if (it != cache.end()) return it->second;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, threads, lds_bytes) != hipSuccess || nb < 1) nb = 0;
    cache[k] = nb;
    return nb;
}
int welch_rp_groups_per_cu(const Xf &xf, bool lin) {
    int res = 0;
#define M_(XT)                                                                                        \
    if constexpr (XT::EXACT && XT::L >= SP_RP_HINT_MIN) {                                             \
        res = lin ? resident_per_cu((const void *)k_welch_rp<XT, true>, XT::C::WG, XT::C::lds_bytes(1))   \
                  : resident_per_cu((const void *)k_welch_rp_h<XT, false>, XT::C::WG, XT::C::lds_bytes(1)); \
    } else {                                                                                          \
        res = lin ? resident_per_cu((const void *)k_welch_rp<XT, true>, XT::C::WG, XT::C::lds_bytes(1))   \
                  : resident_per_cu((const void *)k_welch_rp<XT, false>, XT::C::WG, XT::C::lds_bytes(1)); \
    }
    SP_DISPATCH_X(xf, M_)
#undef M_
    // 1 or 2 resident: 4 per CU as before (whole rounds); 3: 3; more: one round
    if (res <= 0) return default_groups_per_cu();
    if (res <= 2) return 4;
    return res > 8 ? 8 : res;
}
// real input, two frames per transform (power only); rp partitions PAIRS of frames
int launch_welch_rp(LaunchCtx c, const float *x, const float *win, int hop, int64_t nframes, const float *trend, bool lin,
                    const Xf &xf, float *partial, const RunPart &rp) {
#define M_(XT)                                                                                        \
    if constexpr (XT::EXACT && XT::L >= SP_RP_HINT_MIN) {                                             \
        /* (not for the linear-detrend variants: the hint makes them spill) */                        \
        if (lin) hipLaunchKernelGGL((k_welch_rp<XT, true>), dim3(rp.blocks), dim3(XT::C::WG), XT::C::lds_bytes(1), c.stream, \
                                    x, win, hop, nframes, rp.fpg, trend, xf.tb, partial);             \
        else hipLaunchKernelGGL((k_welch_rp_h<XT, false>), dim3(rp.blocks), dim3(XT::C::WG), XT::C::lds_bytes(1), c.stream, \
                                x, win, hop, nframes, rp.fpg, trend, xf.tb, partial);                 \
    } else {                                                                                          \
        if (lin) hipLaunchKernelGGL((k_welch_rp<XT, true>), dim3(rp.blocks), dim3(XT::C::WG), XT::C::lds_bytes(1), c.stream, \
                                    x, win, hop, nframes, rp.fpg, trend, xf.tb, partial);             \
        else hipLaunchKernelGGL((k_welch_rp<XT, false>), dim3(rp.blocks), dim3(XT::C::WG), XT::C::lds_bytes(1), c.stream, \
                                x, win, hop, nframes, rp.fpg, trend, xf.tb, partial);                 \
    }
    SP_DISPATCH_X(xf, M_)
#undef M_
    return 0;
}

// ---- one-pass detrend epilogue: 4 launches (estimate | main kernel | two reductions | totals) + 1 at finish -------
int launch_op_estimate(LaunchCtx c, const void *x, bool cplx, int64_t nsig, double *, float *trend) {
    if (cplx) hipLaunchKernelGGL((k_op_estimate<true>), dim3(1), dim3(1024), 0, c.stream, x, nsig, trend);
    else hipLaunchKernelGGL((k_op_estimate<false>), dim3(1), dim3(1024), 0, c.stream, x, nsig, trend);
    return 0;
}

// A[k] raw sums, Sl[j] block sums, tot = sum_{i<nmean}(x - mu0), local delta, plain sample sum
int launch_op_reduce(LaunchCtx c, const void *x, bool cplx, const float *trend, const float *partial, const cf *spartial,
                     int64_t G, const Xf &xf, int hop, int64_t nframes, int64_t nmean, OnePass st, double *sum_out) {
    const int N = xf.L, H = hop, r = N / H;
    hipLaunchKernelGGL(k_op_colsums, dim3((N + 31) / 32 + (2 * H + 31) / 32), dim3(1024), 0, c.stream, partial, N, st.A,
                       reinterpret_cast<const float *>(spartial), 2 * H, st.Sl, G);
    if (!sum_out) return 0;          // single-call path: k_op_finish derives the shard mean itself
    if (cplx) hipLaunchKernelGGL((k_op_total<true>), dim3(1), dim3(1024), 0, c.stream, x, trend, st.Sl, H, r, nframes, nmean, st.tot, st.dlt, sum_out);
    else hipLaunchKernelGGL((k_op_total<false>), dim3(1), dim3(1024), 0, c.stream, x, trend, st.Sl, H, r, nframes, nmean, st.tot, st.dlt, sum_out);
    return 0;
}

// c -> B = FFT(w c) -> combine, one workgroup.  export_state: write the shard's additive state instead (out = state)
int launch_op_finish(LaunchCtx c, const void *x, bool cplx, const float *trend, const float *win, OnePass st,
                     const double *mean_in, int64_t nmean, const Xf &xf, int hop, int64_t nframes, cf *, const cf *Wf,
                     int sided, double scale, double *out, bool export_state) {
    const int H = hop, r = xf.L / H;
#define FIN_(NN)                                                                                      \
    case NN:                                                                                          \
        if (export_state) {                                                                           \
            if (cplx)                                                                                 \
                hipLaunchKernelGGL((k_op_finish<NN, true, true>), dim3(1), dim3(WgCfg<NN>::WG), WgCfg<NN>::lds_bytes(1), c.stream, \
                                   x, trend, win, st.Sl, st.A, Wf, st.dlt, mean_in, H, r, nframes, nmean, sided, scale, xf.tb, out, st.sym); \
            else                                                                                      \
                hipLaunchKernelGGL((k_op_finish<NN, false, true>), dim3(1), dim3(WgCfg<NN>::WG), WgCfg<NN>::lds_bytes(1), c.stream, \
                                   x, trend, win, st.Sl, st.A, Wf, st.dlt, mean_in, H, r, nframes, nmean, sided, scale, xf.tb, out, st.sym); \
        } else if (cplx)                                                                              \
            hipLaunchKernelGGL((k_op_finish<NN, true>), dim3(1), dim3(WgCfg<NN>::WG), WgCfg<NN>::lds_bytes(1), c.stream, x, \
                               trend, win, st.Sl, st.A, Wf, st.dlt, mean_in, H, r, nframes, nmean, sided, scale, xf.tb, out, st.sym); \
        else                                                                                          \
            hipLaunchKernelGGL((k_op_finish<NN, false>), dim3(1), dim3(WgCfg<NN>::WG), WgCfg<NN>::lds_bytes(1), c.stream, x, \
                               trend, win, st.Sl, st.A, Wf, st.dlt, mean_in, H, r, nframes, nmean, sided, scale, xf.tb, out, st.sym); \
        break;
    switch (xf.L) {
        FIN_(256) FIN_(512) FIN_(1024) FIN_(2048) FIN_(4096) FIN_(8192)
        default: return -1;
    }
#undef FIN_
    return 0;
}

int launch_op_fused(LaunchCtx c, const void *x, bool cplx, const float *trend, const float *win, const float *partial,
                    const cf *spartial, int64_t G, int n, int hop, int64_t nframes, int64_t nmean, OnePass st, unsigned *ticket,
                    CogLobe lb, const double *mean_in, int sided, double scale, double *out, bool export_state, OpPrev prev, bool light,
                    double cola_c) {
    if (n % hop != 0 || n / hop > 4 || (hop & (hop - 1)) != 0 || n % 32 != 0 || (2 * hop) % 32 != 0 || lb.K < 0 || lb.K > 3 || n < 2 * lb.K + 2)
        return -1;
    const bool lobeb = cola_c > 0.0;             // the main kernel left lobe sums (k_welch_pipe mode 9), not block sums
    const int wg = light ? SP_OPF_WG_LIGHT : SP_OPF_WG;
    const dim3 grid(n / 32 + (lobeb ? 0 : (2 * hop) / 32)), block(wg);
    const double ang = -2.0 * M_PI * (double)wg / (double)n;                   // the bin-to-bin rotation of the last block's twiddles
    const double step_c = cos(ang), step_s = sin(ang);
#define FUSED_(CP, EX, EE, LT, LB)                                                                     \
    hipLaunchKernelGGL((k_op_fused<CP, EX, EE, LT, LB>), grid, block, 0, c.stream, partial, n, st.A, reinterpret_cast<const float *>(spartial), \
                       hop, st.Sl, G, ticket, x, trend, win, lb, mean_in, nframes, nmean, sided, scale, out, st.sym, prev, step_c, step_s, cola_c)
#define FUSED_E_(CP, EX)                                                                               \
    {                                                                                                 \
        if (lobeb) {                                                                                  \
            if (light) FUSED_(CP, EX, 1, true, true); else FUSED_(CP, EX, 1, false, true);             \
        } else if (light) {                                                                           \
            if (n / hop <= 2) FUSED_(CP, EX, 1, true, false); else FUSED_(CP, EX, 3, true, false);     \
        } else {                                                                                      \
            if (n / hop <= 2) FUSED_(CP, EX, 1, false, false); else FUSED_(CP, EX, 3, false, false);   \
        }                                                                                             \
    }
    if (cplx) {
        if (export_state) FUSED_E_(true, true) else FUSED_E_(true, false)
    } else {
        if (export_state) FUSED_E_(false, true) else FUSED_E_(false, false)
    }
#undef FUSED_E_
#undef FUSED_
    return 0;
}

int launch_op_apply(LaunchCtx c, const double *state, const cf *Wf, int n, int sided, double scale, double *out) {
    hipLaunchKernelGGL(k_op_apply, dim3((n + 255) / 256), dim3(256), 0, c.stream, state, Wf, n, sided, scale, out);
    return 0;
}

// one-pass mean detrend of MANY real channels (the CSD matrix on packed pair spectra): per channel, from the block sums Sl and
// the mean estimate in its trend record, the state of k_op_finish<EXPORT>: B = sum_g X_g, the plain sample sum, ... (5 L + 8
// doubles per channel)
int launch_op_finish_channels(LaunchCtx c, const void *x, int64_t x_cs, int nch, const float *trend, const float *win,
                              const double *Sl, const cf *Wf, int hop, int64_t nframes, int64_t nmean, const Xf &xf, double *out,
                              bool cplx) {
    if (xf.L != 4096) return -1;
    const int H = hop, r = xf.L / H;
    if (cplx) {
        hipLaunchKernelGGL((k_op_finish<4096, true, true>), dim3(nch), dim3(WgCfg<4096>::WG), WgCfg<4096>::lds_bytes(1), c.stream, x,
                           trend, win, Sl, Sl, Wf, (const double *)nullptr, (const double *)nullptr, H, r, nframes, nmean, 2, 1.0, xf.tb,
                           out, 0, x_cs, (int64_t)2 * H, (int64_t)5 * xf.L + 8);
        return 0;
    }
    hipLaunchKernelGGL((k_op_finish<4096, false, true>), dim3(nch), dim3(WgCfg<4096>::WG), WgCfg<4096>::lds_bytes(1), c.stream,
                       x, trend, win, Sl, Sl /* A: unused by the consumers of this state */, Wf, (const double *)nullptr,
                       (const double *)nullptr, H, r, nframes, nmean, 2, 1.0, xf.tb, out, 0, x_cs, (int64_t)2 * H,
                       (int64_t)5 * xf.L + 8);
    return 0;
}

}   // namespace sp
