// k_welch_pipe.hip -- the metric shape of the one-pass Welch PSD as a three-stage pipeline of specialised waves.
//
// k_welch_carry (kernels.h) gives every wave the whole frame: 601 VALU instructions issued at the single-wave rate, two LDS
// exchanges, 205 VGPRs = 2 waves per SIMD, and the SIMD's VALU pipe idles whenever one of its two waves is exchanging
// (DESIGN.md section 4).  Here a workgroup is 12 waves = 3 roles x 4 waves, one wave of each role per SIMD, and a frame
// moves through the roles in three consecutive periods:
//   role 0 (front):  stream the hop's new samples (overlap carried in registers, one frame prefetched), detrend by mu0,
//                    block sums of the one-pass mean correction, window, radix-16 pass 0, scatter into image A[p & 1]
//   role 1 (middle): gather A (into registers, one period ahead of use), twiddled radix-16 pass 1, scatter into image B
//   role 2 (back):   gather B (one period ahead of use), twiddled radix-16 pass 2, |X|^2 accumulated in 16 registers
// with ONE workgroup barrier per period (both images double-buffered: 4 x 32.1 KiB of LDS, one workgroup per CU).  Each
// role keeps only its own constants (<= 128 VGPRs), the three waves of a SIMD issue VALU concurrently, and the LDS traffic
// per frame is that of the plain kernel (128 KiB).  Results are bit-identical to k_welch_carry's per group of frames
// (same arithmetic in the same order per frame; the partition into groups differs).
// Reference path: fft_analysis.py:2126-2203 fft_win -> :1946 Pstft -> :1980 averagewins.
#include "launch.h"
#include <type_traits>
// SP_PIPE_ES=1: the 16 scatter stores of a pass leave 4 at a time behind the radix-4 butterfly that produces them
#ifndef SP_PIPE_ES
#define SP_PIPE_ES 1
#endif
// SP_PIPE_AHEAD: the front role loads the new samples 1 or 2 frames ahead of use
// SP_PIPE_SPREAD=1 (with SP_PIPE_AHEAD=2): the front role issues its loads in four groups spread over the period.  Defaults 2 / 1:
// -2 % per bench step (0.570 against 0.583 ms, three interleaved runs); two frames ahead alone changes nothing
#ifndef SP_PIPE_SPREAD
#define SP_PIPE_SPREAD 1
#endif
// SP_PIPE_NT=1: the front role streams the samples with the non-temporal policy
#ifndef SP_PIPE_NT
#define SP_PIPE_NT 1
#endif
#ifndef SP_PIPE_TIMING
#define SP_PIPE_TIMING 0
#endif
// SP_PIPE_PRIO=abc: s_setprio of the front / middle / back role (0 = leave the default)
#ifndef SP_PIPE_PRIO
#define SP_PIPE_PRIO 0
#endif
#ifndef SP_PIPE_AHEAD
#define SP_PIPE_AHEAD 2
#endif
// SP_PIPE_WINFOLD=1: the front role folds the window into the first radix-4 stage (dft16s_es_win: 16 VALU less per frame)
#ifndef SP_PIPE_WINFOLD
#define SP_PIPE_WINFOLD 1
#endif
// SP_PIPE_EARLYSPREAD=1 (with the window fold): the three later groups of the front role's spread loads leave behind the first-stage
// butterflies instead of behind the last stage's stores: in flight longer before their use -- but bunched into the first third of
// the period, and measured SLOWER (bench step 0.570 against 0.555 ms, three interleaved warm rounds, profiles/r03_pipe_ab.txt):
// what the loads need is even spacing (the CU's miss queue), not more lead (three frames ahead is no faster either).  Default off.
#ifndef SP_PIPE_EARLYSPREAD
#define SP_PIPE_EARLYSPREAD 0
#endif
// SP_PIPE_UPROT=1: the input rotations (1 - i tau) of pass p + 1 are applied by role p to its outputs before the scatter: 30 VALU
// move from the back role (the longest: 206 VALU + its gather) to the front role (which has slack since the window fold and
// the lobe sums), the middle role's count stays (it gives 30 and takes 30)
#ifndef SP_PIPE_UPROT
#define SP_PIPE_UPROT 1
#endif
#ifndef SP_PIPE_RM
#define SP_PIPE_RM 0          // 1: first exchange image [thread][16] (fft_core.h, WgFft RM): 16-byte scatter writes
#endif
namespace sp {

#if !SP_PACKED
template <bool CPLX, int SHIFT, int MODE>      // MODE 0: plain accumulation, 1: one-pass mean detrend, 2: moments per frame (cog),
                                               // 3 / 4: the same as 0 / 1 for real input with two frames per transform
__global__ __launch_bounds__(768) void k_welch_pipe(const void *__restrict__ x_in, const float *__restrict__ win,
                                                     int64_t nframes, int64_t fpg, float *__restrict__ trend_in, XfTables tb,
                                                     float *__restrict__ partial, cf *__restrict__ spartial, int64_t x_cs, int gpr) {
    constexpr int N = 4096;
    constexpr bool RP = MODE >= 3 && MODE != 8 && MODE != 9;         // real input, two frames per transform (modes 3: plain, 4: one-pass detrend, 5: spectra)
    constexpr bool ONEPASS = MODE == 1 || MODE == 4 || MODE == 7 || MODE == 8 || MODE == 9, COG = MODE == 2 || MODE == 8;   // (7: mode 5, 8: mode 2, with the one-pass block sums)
    // mode 9 = mode 1 for windows whose spectrum is confined to the bins -3 .. 3 AND that add up to a constant at this hop
    // (COLA: Hann, Hamming at 50 / 75 % ...): what the epilogue needs of sum_g X_g are those 7 bins, which the BACK role has in
    // registers anyway (bins 0..3 in slot 0 of threads 0..3, bins N-3..N-1 in slot 15 of threads 253..255): 4 additions per
    // frame there instead of the front role's 16 for the block sums, no block sums written, and the signal's plain sum
    // follows from the DC bin, sum_g X_g[0] = sum_i cov(i) (x[i] - mu0) with cov = c but for the two edges (k_op_fused<LOBEB>).
    // `spartial` receives lobeB[group][8] (ks = -3 .. 3 at index ks + 3) instead of the block sums.
    constexpr bool LOBE = MODE == 9;
    // mode 5: no accumulation -- the packed pair spectrum Z = X_2q + i X_2q+1 of every frame pair is WRITTEN, all N bins, for the
    // CSD-matrix contraction: Zs[pair of pairs][group of 8 bins][channel slot of 64][8 bins][2 pairs] (k_csdm_bf16's layout with
    // "frames" = pairs).  The contraction of the PACKED spectra, H[k] = sum Z_i[k] conj Z_j[k], gives the matrix by the mirror
    // combination G[k] = (H[k] + conj H[N-k]) / 2 taken ONCE on the sums (k_csdm_fold) -- no mirror exchange per transform.  The
    // back role keeps the spectrum of an even pair for one period and writes it with the next one as 16-byte stores (full
    // 128-byte lines per 8 lanes).  blockIdx.y = channel (x_cs samples apart, trend record 4 y), gpr = N / 8 bin groups;
    // a workgroup's run starts at an even pair.
    constexpr bool SPEC = MODE == 5 || MODE == 7;
    const void *x = SPEC ? (const void *)(reinterpret_cast<const float *>(x_in) + (int64_t)blockIdx.y * x_cs) : x_in;
    float *trend = SPEC ? trend_in + 4 * blockIdx.y : trend_in;
    static_assert(!RP || (!CPLX && SHIFT == 8), "the real-pair form is for real input at hop = nfft / 2");
    using PL = FftPlan<N>;
    using F = WgFft<N, false, SP_PIPE_RM != 0>;
    constexpr int T = PL::T, R = PL::R, KEEP = R - SHIFT, IMG = F::IMG0, IMGB = SP_PIPE_RM ? N : PL::LDS_ELEMS;
    static_assert(T == 256 && R == 16 && PL::NP == 3, "three radix-16 passes over 256 threads");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *smem = reinterpret_cast<cf *>(smem_raw);
    cf *imgA = smem, *imgB = smem + 2 * IMG;
    const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
    const int tid = (int)threadIdx.x & 255;
    const int hop = SHIFT * T;

    // mu0: the same 16 runs of 256 samples in every workgroup (identical order -> identical value everywhere); any value
    // gives the exact result, the epilogue corrects with the true mean.  Workgroup 0 publishes it for the epilogue kernels.
    cf mu;
    if constexpr (ONEPASS) {
        const int64_t span = (nframes - 1) * (int64_t)hop + N;
        const int64_t pitch = span / 16;
        // SPEC (many channels, cross terms): 16 runs of 3072 samples read by all twelve waves.  The spectra carry d W with
        // d = mean - mu0 in the bins of the window's main lobe, coherently in every frame and channel, until the epilogue
        // removes it again: the float32 accumulators of the contraction lose what that sum exceeds the signal by, so the
        // estimate has to be good (d ~ sigma / 220; with 4096 samples the full-size Hermitian check saw 7e-6 of the peak)
        constexpr int NU = SPEC ? 4 : 1, NTH = SPEC ? 3 * T : T, NWV = SPEC ? 12 : 4;
        const int lt = SPEC ? (int)threadIdx.x : tid;
        float sx = 0.f, sy = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                int64_t i = pitch * r + (int64_t)(lt + NTH * u);
                i = i < span ? i : span - 1;
                const cf v = load_sample(x, i, CPLX);
                sx += v.x;
                sy += v.y;
            }
        sx = wave_sum64(sx);
        sy = wave_sum64(sy);
        float *red = reinterpret_cast<float *>(smem);
        if ((SPEC || role == 0) && (tid & 63) == 0) {
            red[2 * (lt >> 6)] = sx;
            red[2 * (lt >> 6) + 1] = sy;
        }
        __syncthreads();
        double tx = 0.0, ty = 0.0;
#pragma unroll
        for (int wv = 0; wv < NWV; ++wv) {
            tx += (double)red[2 * wv];
            ty += (double)red[2 * wv + 1];
        }
        mu = mk((float)(tx / (16.0 * NTH * NU)), (float)(ty / (16.0 * NTH * NU)));
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            trend[0] = mu.x;
            trend[1] = mu.y;
            trend[2] = 0.f;
            trend[3] = 0.f;
        }
    } else {
        mu = load_trend(trend).m;          // the caller's constant (or zero): plain accumulation, no epilogue
    }
    const int64_t gid = blockIdx.x;
    const int64_t g0 = gid * fpg;                        // first frame (RP: first frame PAIR) of this workgroup
    const int64_t last = nframes - 1;
    int64_t trips = (RP ? (nframes + 1) / 2 : nframes) - g0;
    trips = trips < 0 ? 0 : (trips > fpg ? fpg : trips);
#if SP_PIPE_TIMING
    // diagnostic: per role, cycles between leaving a barrier and arriving at the next one (busy) and cycles spent at the barrier
    unsigned long long t_busy = 0, t_wait = 0, t_issue = 0, t_mark = __builtin_amdgcn_s_memtime();
#define PIPE_SYNC()                                                                                   \
    {                                                                                                 \
        const unsigned long long ta_ = __builtin_amdgcn_s_memtime();                                  \
        __syncthreads();                                                                              \
        const unsigned long long tb_ = __builtin_amdgcn_s_memtime();                                  \
        t_busy += ta_ - t_mark;                                                                       \
        t_wait += tb_ - ta_;                                                                          \
        t_mark = tb_;                                                                                 \
    }
#else
#define PIPE_SYNC() __syncthreads()
#endif
    constexpr int DRAIN = 4;                 // a frame leaves the pipeline 4 periods after it entered
    const int64_t periods = trips + DRAIN;
    F f;
    // (not for the real-pair fronts: three rotating register sets + window leave no room for 16 more constants -- 12-37 spills)
    constexpr bool UPROT = SP_PIPE_UPROT && !RP && !SP_PIPE_RM && !SP_ABLATE && SP_PIPE_ES;
    float tau_out[16];                       // UPROT: rotations this role applies to its outputs (front: for pass 1, middle: for pass 2)
    if constexpr (UPROT) {
        if (role == 0) f.template load_tau_out<0>(tb.tw, tid, tau_out);
        else if (role == 1) f.template load_tau_out<1>(tb.tw, tid, tau_out);
    }

    if (role == 0 && RP) {
        // real input, frames 2q and 2q + 1 in one transform: z = f_2q + i f_2q+1 (the finish kernel symmetrises |Z|^2).  At hop
        // = N/2 the pair q needs the chunk q = samples [q N, q N + N) as its real part and [second half of chunk q | first half
        // of chunk q + 1] as its imaginary part: every sample is loaded once, 16 floats per thread and period, one period
        // before it is first used (three register sets rotate: current, next, incoming)
        const float *xr = reinterpret_cast<const float *>(x);
        float m = mu.x;
        asm volatile("" : "+v"(m));
        float w[R], sacc[SHIFT];
#pragma unroll
        for (int t = 0; t < R; ++t) w[t] = win[tid + T * t];
#pragma unroll
        for (int s = 0; s < SHIFT; ++s) sacc[s] = 0.f;
        // chunk q, clamped per half to a half that exists (first half: 2q <= nframes, second: 2q + 1 <= nframes)
        auto issue_chunk = [&](float (&dst)[R], int64_t q) __attribute__((always_inline)) {
            const int64_t q1 = 2 * q <= nframes ? q : 0, q2 = 2 * q + 1 <= nframes ? q : 0;
            const float *b1 = xr + q1 * (int64_t)N, *b2 = xr + q2 * (int64_t)N;
#pragma unroll
            for (int t = 0; t < R; ++t) {
                const unsigned off = (unsigned)(tid + T * t);
#if SP_PIPE_NT
                dst[t] = __builtin_nontemporal_load((t < R / 2 ? b1 : b2) + off);
#else
                dst[t] = (t < R / 2 ? b1 : b2)[off];
#endif
            }
        };
        float ca[R], cb[R], cc[R];
        issue_chunk(ca, g0);
        issue_chunk(cb, g0 + 1);
#pragma unroll
        for (int t = 0; t < R; ++t) {
            ca[t] -= m;
            cb[t] -= m;
        }
        // hb: frame 2q + 1 exists (false only for the last pair of an odd frame count, which the tail loop below handles)
        auto frame = [&](auto hb, int64_t i, cf *img, float (&cur)[R], float (&nxt)[R], float (&fill)[R]) __attribute__((always_inline)) {
            constexpr bool HB = decltype(hb)::value;
            issue_chunk(fill, g0 + i + 2);
            __builtin_amdgcn_sched_barrier(0);
            cf v[R];
            // (not with the one-pass block sums: modes 4 / 7 then spill 17-20 registers instead of 2-6)
            if constexpr (SP_PIPE_WINFOLD && !ONEPASS && !SP_PIPE_RM && !SP_ABLATE) {
#pragma unroll
                for (int t = 0; t < R; ++t) v[t] = mk(cur[t], HB ? (t < R / 2 ? cur[t + R / 2] : nxt[t - R / 2]) : 0.f);
                if constexpr (UPROT) f.bfly_scatter_win_rot(v, w, tau_out, img, tid);
                else f.bfly_scatter_win(v, w, img, tid);
            } else {
#pragma unroll
                for (int t = 0; t < R; ++t) v[t] = mk(w[t] * cur[t], HB ? w[t] * (t < R / 2 ? cur[t + R / 2] : nxt[t - R / 2]) : 0.f);
                if constexpr (UPROT) f.template bfly_scatter_rot<0, false>(v, tau_out, img, tid);
                else f.template bfly_scatter<0>(v, img, tid);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ONEPASS) {
                // last hop-block of frame 2q (second half of the chunk) and of frame 2q + 1 (first half of the next one)
#pragma unroll
                for (int s = 0; s < SHIFT; ++s) sacc[s] += HB ? cur[R / 2 + s] + nxt[s] : cur[R / 2 + s];
            }
#pragma unroll
            for (int t = 0; t < R; ++t) fill[t] -= m;
        };
        const std::true_type yes;
        const std::false_type no;
        int64_t i = 0;
        for (; i + 6 < trips; i += 6) {                           // (never the last pair: the tail loop owns it)
            frame(yes, i, imgA, ca, cb, cc);
            PIPE_SYNC();
            frame(yes, i + 1, imgA + IMG, cb, cc, ca);
            PIPE_SYNC();
            frame(yes, i + 2, imgA, cc, ca, cb);
            PIPE_SYNC();
            frame(yes, i + 3, imgA + IMG, ca, cb, cc);
            PIPE_SYNC();
            frame(yes, i + 4, imgA, cb, cc, ca);
            PIPE_SYNC();
            frame(yes, i + 5, imgA + IMG, cc, ca, cb);
            PIPE_SYNC();
        }
        for (; i < trips; ++i) {                                  // i is a multiple of 6 at entry: the rotation state is i % 3
            cf *img = (i & 1) ? imgA + IMG : imgA;
            const bool has_b = 2 * (g0 + i) + 1 < nframes;        // uniform
            if (has_b) {
                if (i % 3 == 0) frame(yes, i, img, ca, cb, cc);
                else if (i % 3 == 1) frame(yes, i, img, cb, cc, ca);
                else frame(yes, i, img, cc, ca, cb);
            } else {
                if (i % 3 == 0) frame(no, i, img, ca, cb, cc);
                else if (i % 3 == 1) frame(no, i, img, cb, cc, ca);
                else frame(no, i, img, cc, ca, cb);
            }
            PIPE_SYNC();
        }
#pragma unroll
        for (int d = 0; d < DRAIN; ++d) PIPE_SYNC();
        if constexpr (ONEPASS) {
#pragma unroll
            for (int s = 0; s < SHIFT; ++s)
                spartial[((SPEC ? (int64_t)blockIdx.y * gridDim.x : 0) + gid) * hop + tid + T * s] = mk(sacc[s], 0.f);   // SPEC: [channel][run][hop]
        }
    } else if (role == 0) {
        if constexpr (SP_PIPE_PRIO) __builtin_amdgcn_s_setprio((SP_PIPE_PRIO / 100) % 10);
        // keep the constant in VGPRs (an SGPR source halves the VALU issue rate on gfx950)
        asm volatile("" : "+v"(mu.x), "+v"(mu.y));
        float w[R];
        cf sacc[SHIFT], raw[R];
#pragma unroll
        for (int t = 0; t < R; ++t) w[t] = win[tid + T * t];
#pragma unroll
        for (int s = 0; s < SHIFT; ++s) sacc[s] = mk(0.f, 0.f);
        {
            const int64_t base = (g0 < nframes ? g0 : last) * hop + tid;
#pragma unroll
            for (int t = 0; t < R; ++t) raw[t] = load_sample(x, base + T * t, CPLX) - mu;
        }
        // new slots of frame g0 + q (clamped at the end of the signal; unused then): scalar base + lane offset
        // slots [s0, s1) of the new samples of frame g0 + q
        auto issue_part = [&](cf (&dst)[SHIFT], int64_t q, int s0, int s1) __attribute__((always_inline)) {
            const int64_t gq = g0 + q;
            int64_t gn = gq < nframes ? gq : last;
            if constexpr (SP_ABLATE & 16) gn = (blockIdx.x & 7) + (q & 1);          // diagnostic: every load hits L2
            const int64_t ubase = gn * hop + (int64_t)T * KEEP;
#pragma unroll
            for (int s = s0; s < s1; ++s) {
                const unsigned off = (unsigned)(tid + T * s);
                if constexpr (SP_ABLATE & 8) {          // diagnostic: no global loads in the loop
                    dst[s] = raw[s] + mu;
                    continue;
                }
#if SP_PIPE_NT
                if (CPLX) {
                    const sp_f2v r = __builtin_nontemporal_load(reinterpret_cast<const sp_f2v *>(x) + ubase + off);
                    dst[s] = mk(r.x, r.y);
                } else {
                    dst[s] = mk(__builtin_nontemporal_load(reinterpret_cast<const float *>(x) + ubase + off), 0.f);
                }
#else
                if (CPLX) dst[s] = (reinterpret_cast<const cf *>(x) + ubase)[off];
                else dst[s] = mk((reinterpret_cast<const float *>(x) + ubase)[off], 0.f);
#endif
            }
        };
        auto issue = [&](cf (&dst)[SHIFT], int64_t q) __attribute__((always_inline)) { issue_part(dst, q, 0, SHIFT); };
        // two frames ahead + spread loads up to hop = nfft/2; at hop = nfft (16 new slots per frame) a second set of incoming
        // registers does not fit (56 spills): one frame ahead, one burst
        // (SP_PIPE_AHEAD=3: three frames ahead -- the new slots are then in flight for two full periods before their first use;
        //  with two, the groups issued late in a period (SPREAD) have little more than one, about the loaded HBM latency.
        //  Round 3: not usable as written -- the six-fold unrolled rotation of three register sets compiles to 168 VGPRs with
        //  47 spilled; kept behind the knob)
        constexpr int AHEAD = (SP_PIPE_AHEAD >= 2 && SHIFT <= 8) ? (SP_PIPE_AHEAD >= 3 ? 3 : 2) : 1;
        constexpr bool SPREAD = SP_PIPE_SPREAD && AHEAD >= 2 && !SP_ABLATE;
        // one period: loads of frame i + AHEAD go out first and are consumed AHEAD periods later (`fill`); `take` holds the
        // new slots of frame i + 1
        auto frame = [&](int64_t i, cf *img, cf (&fill)[SHIFT], cf (&take)[SHIFT]) __attribute__((always_inline)) {
#if SP_PIPE_TIMING
            const unsigned long long ti0_ = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
#endif
            if constexpr (SPREAD) issue_part(fill, i + AHEAD, 0, SHIFT / 4);
            else issue(fill, i + AHEAD);
            __builtin_amdgcn_sched_barrier(0);          // keep the loads above the arithmetic (hipcc sank them to the barrier)
#if SP_PIPE_TIMING
            t_issue += __builtin_amdgcn_s_memtime() - ti0_;
            __builtin_amdgcn_sched_barrier(0);
#endif
            cf v[R];
            constexpr bool WINFOLD = SP_PIPE_WINFOLD && SPREAD && !SP_PIPE_RM;
#pragma unroll
            for (int t = 0; t < R; ++t) v[t] = WINFOLD ? raw[t] : w[t] * raw[t];
            if constexpr (SPREAD) {
            // the other three quarters of the loads leave behind the store groups of the butterfly: a CU keeps about
            // 24-32 KiB of misses in flight (tools/ubench/stream_mlp.hip) and HBM latency is about one period, so a burst of
            // 16 KiB at the top of a period blocks at issue (350 cycles per period measured) while the same loads spread
            // over the period find the queue drained
            {
                constexpr bool EARLY = SP_PIPE_EARLYSPREAD && WINFOLD;
                auto store = [&](int k, cf val) __attribute__((always_inline)) {
                    img[F::template phys<0>(tid * 16 + k)] = UPROT ? rot_tan(val, tau_out[k]) : val;
                    if constexpr (!EARLY)
                        if (k >= 12 && k < 15) issue_part(fill, i + AHEAD, (k - 11) * (SHIFT / 4), (k - 10) * (SHIFT / 4));
                };
                auto mid = [&](int b) __attribute__((always_inline)) { issue_part(fill, i + AHEAD, (b + 1) * (SHIFT / 4), (b + 2) * (SHIFT / 4)); };
                if constexpr (EARLY) dft16s_es_win(v, w, store, mid);
                else if constexpr (WINFOLD) dft16s_es_win(v, w, store);
                else dft16s_es<false>(v, f.t16[0], store);
            }
            } else {
#if SP_PIPE_ES && !SP_ABLATE
            if constexpr (UPROT) f.template bfly_scatter_rot<0, false>(v, tau_out, img, tid);
            else f.template bfly_scatter<0>(v, img, tid);
#else
            f.template bfly<0>(v, tid);
            if constexpr (!(SP_ABLATE & 2)) f.template scatter<0>(v, img, tid);
            else asm volatile("" ::"v"(v[0].x), "v"(v[5].y), "v"(v[10].x), "v"(v[15].y));
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
            // off the critical path of the period: the stores above drain while these issue
#pragma unroll
            for (int s = 0; s < SHIFT; ++s)
                if constexpr (ONEPASS && !LOBE) sacc[s] = sacc[s] + raw[KEEP + s];
#pragma unroll
            for (int t = 0; t < KEEP; ++t) raw[t] = raw[t + SHIFT];
#pragma unroll
            for (int s = 0; s < SHIFT; ++s) raw[KEEP + s] = take[s] - mu;
        };
        int64_t i = 0;
        if constexpr (AHEAD == 3) {
            // three rotating sets: at period i `fill` = S[i % 3] receives frame i + 3, `take` = S[(i + 1) % 3] holds frame i + 1
            cf s0[SHIFT], s1[SHIFT], s2[SHIFT];
            issue(s1, 1);
            issue(s2, 2);
            for (; i + 2 < trips; i += 3) {            // (image by parity of the period: a scalar select, so that the loop unrolls by 3 only)
                frame(i, (i & 1) ? imgA + IMG : imgA, s0, s1);
                PIPE_SYNC();
                frame(i + 1, ((i + 1) & 1) ? imgA + IMG : imgA, s1, s2);
                PIPE_SYNC();
                frame(i + 2, ((i + 2) & 1) ? imgA + IMG : imgA, s2, s0);
                PIPE_SYNC();
            }
            for (; i < trips; ++i) {                   // i is a multiple of 3 at entry of this tail: rotation state i % 3
                cf *img = (i & 1) ? imgA + IMG : imgA;
                if (i % 3 == 0) frame(i, img, s0, s1);
                else if (i % 3 == 1) frame(i, img, s1, s2);
                else frame(i, img, s2, s0);
                PIPE_SYNC();
            }
        } else if constexpr (AHEAD == 2) {
            cf nxa[SHIFT], nxb[SHIFT];
            issue(nxa, 1);
            for (; i + 1 < trips; i += 2) {
                frame(i, imgA, nxb, nxa);
                PIPE_SYNC();
                frame(i + 1, imgA + IMG, nxa, nxb);
                PIPE_SYNC();
            }
            if (i < trips) {
                frame(i, imgA, nxb, nxa);
                PIPE_SYNC();
            }
        } else {
            for (; i + 1 < trips; i += 2) {
                cf nx0[SHIFT], nx1[SHIFT];
                frame(i, imgA, nx0, nx0);
                PIPE_SYNC();
                frame(i + 1, imgA + IMG, nx1, nx1);
                PIPE_SYNC();
            }
            if (i < trips) {
                cf nx0[SHIFT];
                frame(i, imgA, nx0, nx0);
                PIPE_SYNC();
            }
        }
#pragma unroll
        for (int d = 0; d < DRAIN; ++d) PIPE_SYNC();
#pragma unroll
        for (int s = 0; s < SHIFT; ++s)
            if constexpr (ONEPASS && !LOBE) spartial[gid * hop + tid + T * s] = sacc[s];
    } else if (role == 1) {
        if constexpr (SP_PIPE_PRIO) __builtin_amdgcn_s_setprio((SP_PIPE_PRIO / 10) % 10);
        // period p: the gather of frame p-1 is ISSUED first and lands while the butterflies of frame p-2 (gathered one period
        // earlier into the other register set) run -- no wave starts a period by waiting for the LDS pipe
        f.template load_tw_one<1>(tb.tw, tid);
        cf va[R], vb[R];
        auto step = [&](cf (&fill)[R], cf (&use)[R], int64_t p, const cf *src, cf *dst) __attribute__((always_inline)) {
            if (p >= 1 && p <= trips) {
                if constexpr (!(SP_ABLATE & 2)) f.template gather<0>(fill, src, tid);
                else {
#pragma unroll
                    for (int t = 0; t < R; ++t) fill[t] = mk(use[t].y + 1.f, use[t].x);
                }
            }
            if (p >= 2 && p <= trips + 1) {
#if SP_PIPE_ES && !SP_ABLATE
                if constexpr (UPROT) f.template bfly_scatter_rot<1, true>(use, tau_out, dst, tid);
                else f.template bfly_scatter<1>(use, dst, tid);
#else
                f.template bfly<1>(use, tid);
                if constexpr (!(SP_ABLATE & 2)) f.template scatter<1>(use, dst, tid);
                else asm volatile("" ::"v"(use[0].x), "v"(use[5].y), "v"(use[10].x), "v"(use[15].y));
#endif
            }
            PIPE_SYNC();
        };
        for (int64_t p = 0; p < periods; p += 2) {
            step(va, vb, p, imgA + IMG, imgB);
            if (p + 1 < periods) step(vb, va, p + 1, imgA, imgB + IMGB);
        }
    } else {
        if constexpr (SP_PIPE_PRIO) __builtin_amdgcn_s_setprio(SP_PIPE_PRIO % 10);
        f.template load_tw_one<2>(tb.tw, tid);
        float acc[R];
#pragma unroll
        for (int t = 0; t < R; ++t) acc[t] = 0.f;
        cf bs_lo = mk(0.f, 0.f), bs_hi = mk(0.f, 0.f);          // LOBE: sum over the frames of bin tid (slot 0) and bin tid + 15 T (slot 15)
        // SPEC: offset (16-byte units) of bin k = tid + T t inside a pair-of-pairs block: (k / 8) 512 + k % 8 = zlo + 16384 t
        const unsigned zlo = (unsigned)((tid >> 3) * 512 + (tid & 7));
        cf held[SPEC ? R : 1];                 // SPEC: the spectrum of the even pair, waiting for its odd partner
        cf va[R], vb[R];
        // COG (Doppler.cog / cogspec, Doppler.py:43-81): instead of summing |X|^2 over the frames, every frame's moments
        // sum ks |X|^2, sum |X|^2 (signed bin index ks, every bin) are reduced across the wave; a wave keeps the moments of its
        // last 64 frames spread over its lanes and writes them with one coalesced store, layout slots[wave][frame] (as
        // k_welch_carry_cog; k_cog_finish adds the four waves of a frame)
        cf pend = mk(0.f, 0.f);
        auto flush = [&](int64_t i) __attribute__((always_inline)) {
            const int sel = (int)(i & 63), li = tid & 63;
            const int64_t gg = g0 + (i - sel) + li;
            if (li <= sel && gg < nframes) reinterpret_cast<cf *>(partial)[(int64_t)(tid >> 6) * nframes + gg] = pend;
        };
        auto step = [&](cf (&fill)[R], cf (&use)[R], int64_t p, const cf *src) __attribute__((always_inline)) {
            if (p >= 3 && p <= trips + 2) {
                if constexpr (!(SP_ABLATE & 2)) f.template gather<1>(fill, src, tid);
                else {
#pragma unroll
                    for (int t = 0; t < R; ++t) fill[t] = mk(use[t].y + 1.f, use[t].x);
                }
            }
            if (p >= 4 && p <= trips + 3) {
                if constexpr (UPROT) f.template bfly_prerot<2>(use, tid);
                else f.template bfly<2>(use, tid);
                if constexpr (SPEC) {
                    const int64_t pr = g0 + (p - 4);                         // global pair index (uniform); runs start even
                    const bool lone = (pr & 1) == 0 && p == trips + 3;       // a last even pair without a partner
                    if ((pr & 1) == 0 && !lone) {
#pragma unroll
                        for (int t = 0; t < R; ++t) held[t] = use[t];
                    } else {
                        // block base (uniform): (pair / 2) * gpr * 512 + channel * 8, in 16-byte units
                        float4 *zb = reinterpret_cast<float4 *>(partial) + ((pr >> 1) * (int64_t)gpr * 512 + (int64_t)blockIdx.y * 8);
                        unsigned bl = zlo;
                        asm volatile("" : "+v"(bl));                         // (keeps hipcc from holding sixteen offsets in registers)
#pragma unroll
                        for (int t = 0; t < R; ++t) {
                            const float4 q = lone ? make_float4(use[t].x, use[t].y, 0.f, 0.f)
                                                  : make_float4(held[t].x, held[t].y, use[t].x, use[t].y);
                            st_stream(zb + (bl + 16384u * t), q);
                        }
                    }
                } else if constexpr (COG) {
                    if constexpr (MODE == 8) {
                        // one-pass mean detrend of a cosine-sum window: the bins ks = -3 .. 3 of every frame go to lobe[frame][ks + 3]
                        // (behind the slot array); k_cog_finish corrects the moments there with the exact mean
                        const int64_t gf = g0 + (p - 4);
                        cf *lobe = reinterpret_cast<cf *>(partial) + 4 * nframes + gf * 8;
                        if (gf < nframes) {
                            if (tid <= 3) lobe[3 + tid] = use[0];
                            if (tid >= T - 3) lobe[3 - (T - tid)] = use[R - 1];
                        }
                    }
                    // ks = tid + c_t with c_t = T t - (N in the upper half): sum ks p = tid sum p + sum c_t p
                    float numc = 0.f, den = 0.f;
#pragma unroll
                    for (int t = 0; t < R; ++t) {
                        const float pw = cnorm(use[t]);
                        numc = fmaf(pw, (float)(T * t - (t >= R / 2 ? N : 0)), numc);
                        den += pw;
                    }
                    float num = fmaf((float)tid, den, numc);
                    num = wave_sum64(num);
                    den = wave_sum64(den);
                    const int64_t i = p - 4;
                    if ((tid & 63) == (int)(i & 63)) pend = mk(num, den);
                    if ((i & 63) == 63 || i == trips - 1) flush(i);
                } else {
#pragma unroll
                    for (int t = 0; t < R; ++t) acc[t] = fmaf(use[t].y, use[t].y, fmaf(use[t].x, use[t].x, acc[t]));
                    if constexpr (LOBE) {
                        bs_lo = bs_lo + use[0];
                        bs_hi = bs_hi + use[R - 1];
                    }
                }
            }
            PIPE_SYNC();
        };
        for (int64_t p = 0; p < periods; p += 2) {
            step(va, vb, p, imgB + IMGB);
            if (p + 1 < periods) step(vb, va, p + 1, imgB);
        }
        if constexpr (!COG && !SPEC) {
#pragma unroll
            for (int t = 0; t < R; ++t) partial[gid * N + tid + T * t] = acc[t];
            if constexpr (LOBE) {
                if (tid <= 3) spartial[gid * 8 + 3 + tid] = bs_lo;
                if (tid >= T - 3) spartial[gid * 8 + 3 - (T - tid)] = bs_hi;
                if (tid == 4) spartial[gid * 8 + 7] = mk(0.f, 0.f);
            }
        }
    }
#if SP_PIPE_TIMING
    if ((blockIdx.x == 3 || blockIdx.x == 200) && tid == 0)
        printf("block %d role %d: busy %llu wait %llu issue %llu cycles over %lld periods\n", (int)blockIdx.x, role, t_busy, t_wait, t_issue, (long long)periods);
    if (blockIdx.x == 3 && (threadIdx.x & 63) == 0) {
        // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
        const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        printf("hwid wave %d role %d: simd %u wave_slot %u cu %u se %u\n", (int)(threadIdx.x >> 6), role, (hw >> 4) & 3, hw & 15, (hw >> 8) & 15, (hw >> 13) & 7);
    }
#endif
#undef PIPE_SYNC
}
#endif

bool welch_pipe_eligible(const Xf &xf, int hop) {
#if SP_PACKED
    (void)xf;
    (void)hop;
    return false;
#else
    return !xf.blue && xf.L == 4096 && (hop == 2048 || hop == 1024 || hop == 4096);
#endif
}

// mode 2: `partial` is the cog slot array [4][nframes] of (num, den) pairs (spartial unused); mode 3: real input, two
// frames per transform (rp partitions frame PAIRS; spartial != null: one-pass block sums as well)
// mode 5: `partial` receives the packed pair spectra of nch channels (x_cs samples apart; trend records 4 floats apart),
// gpr bin groups per row (k_welch_pipe, SPEC)
int launch_welch_pipe(LaunchCtx c, const void *x, bool cplx, const float *win, int hop, int64_t nframes, float *trend,
                      const Xf &xf, float *partial, const RunPart &rp, cf *spartial, int mode, int nch, int64_t x_cs, int gpr) {
#if SP_PACKED
    return -1;
#else
    const size_t lds = SP_PIPE_RM ? sizeof(cf) * 2 * (size_t)(WgFft<4096, false, true>::IMG0 + 4096)
                                  : sizeof(cf) * 4 * (size_t)FftPlan<4096>::LDS_ELEMS;
#define PIPE_(CP, S, OP)                                                                                  \
    {                                                                                                 \
        static bool once = false;                                                                     \
        if (!once) {                                                                                  \
            if (hipFuncSetAttribute((const void *)k_welch_pipe<CP, S, OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
                return -1;                                                                            \
            once = true;                                                                              \
        }                                                                                             \
        if (c.stop)                                                                                   \
            hipExtLaunchKernelGGL((k_welch_pipe<CP, S, OP>), dim3(rp.blocks, gy), dim3(768), lds, c.stream, nullptr, c.stop, 0, x, win, \
                                  nframes, rp.fpg, trend, xf.tb, partial, spartial, x_cs, gpr);       \
        else                                                                                          \
            hipLaunchKernelGGL((k_welch_pipe<CP, S, OP>), dim3(rp.blocks, gy), dim3(768), lds, c.stream, x, win, nframes, rp.fpg, trend, \
                               xf.tb, partial, spartial, x_cs, gpr);                                  \
    }
    const int shift = hop / 256;
    const unsigned gy = mode == 5 ? (unsigned)nch : 1u;
#define PIPE_S_(CP, OP)                                                                               \
    if (shift == 8) PIPE_(CP, 8, OP) else if (shift == 4) PIPE_(CP, 4, OP) else PIPE_(CP, 16, OP)
    if (mode == 5) {
        if (cplx || shift != 8 || nch < 1 || gpr < 1) return -1;
        if (spartial) PIPE_(false, 8, 7) else PIPE_(false, 8, 5)
    } else if (mode == 3) {
        if (cplx || shift != 8) return -1;
        if (spartial) PIPE_(false, 8, 4) else PIPE_(false, 8, 3)
    } else if (mode == 2 && spartial) {
        // moments per frame with the one-pass mean detrend (mode 8): hop 2048 / 1024 only (16 new slots per thread spill)
        if (shift == 8) { if (cplx) PIPE_(true, 8, 8) else PIPE_(false, 8, 8) }
        else if (shift == 4) { if (cplx) PIPE_(true, 4, 8) else PIPE_(false, 4, 8) }
        else return -1;
    } else if (mode == 2) {
        if (cplx) { PIPE_S_(true, 2) } else { PIPE_S_(false, 2) }
    } else if (mode == 9) {
        // mode 1 with the lobe sums in the back role (hop 2048 / 1024 only, as the one-pass front role)
        if (!spartial || shift == 16) return -1;
        if (shift == 8) { if (cplx) PIPE_(true, 8, 9) else PIPE_(false, 8, 9) }
        else { if (cplx) PIPE_(true, 4, 9) else PIPE_(false, 4, 9) }
    } else if (spartial) {
        if (cplx) { PIPE_S_(true, 1) } else { PIPE_S_(false, 1) }
    } else {
        if (cplx) { PIPE_S_(true, 0) } else { PIPE_S_(false, 0) }
    }
#undef PIPE_S_
#undef PIPE_
    return 0;
#endif
}

}   // namespace sp
