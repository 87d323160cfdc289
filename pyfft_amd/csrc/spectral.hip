// spectral.hip -- C ABI (include/spectral.h) over the gfx950 kernels (kernels.h via launch.h).
#include "../../include/spectral.h"
#include "launch.h"
#include "table_cache.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and enums only: the functions are resolved with dlsym (no link-time dependency)
#include <dlfcn.h>
#include <limits.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

using namespace sp;

#define SP_VERSION 106
#define SP_MAX_WG_FFT 8192
#define SP_MAX_BIG_LOG2 26          /* longest multi-pass power-of-two transform: 2^26 points (512 MiB per buffer) */

namespace {

thread_local std::string g_err;

int fail(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}

#define HIPCHK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define LAUNCHCHK(expr)                                                                               \
    do {                                                                                              \
        if ((expr) != 0) return fail("%s: no kernel instantiated for this transform length (%s:%d)", #expr, __FILE__, __LINE__); \
        HIPCHK(hipGetLastError());                                                                    \
    } while (0)

struct Scratch {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + (bytes >> 3) + 4096;
        if (hipMalloc(&p, want) != hipSuccess) return fail("hipMalloc(%zu) failed", want);
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct BlueTab {
    cf *chirp, *bf;
    int L;
};

struct Ctx {
    bool ready = false;
    int device = 0;
    int ncu = 256;
    hipStream_t stream = nullptr;
    std::map<int64_t, cf *> twiddles;   // L -> exp(-2 pi i m/L)
    std::map<int64_t, BlueTab> blue;    // n -> Bluestein tables
    Scratch in0, in1, out0, work, small, trends, onepass, ticket, epi;
    Scratch pend_trend;                       // trend record a pending sp_welch_accum keeps until sp_welch_finish
    Scratch bigA, bigB, bigT, blueA, blueB, longrec;   // long (multi-kernel) paths
    Scratch cmS, cmT, cmG, cmH, cmO;               // CSD matrix: spectra, bin-major spectra, float64 accumulator, packed-spectra sums, one-pass means state
    std::map<int64_t, BigTw> bigtw;           // N -> two-level twiddle tables of the multi-pass FFT
    std::map<int64_t, BlueTab> blue_big;      // n -> chirp[n], FFT_L(chirp*) (unscaled) for multi-pass Bluestein
    std::mutex mu;
    bool profile = false, prof_valid = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_sw = nullptr;
    const char *last_kernel = "";
} g;

LaunchCtx lc() { return LaunchCtx{g.stream, g.ncu}; }

// pending one-pass Welch accumulation (sp_welch_accum -> sp_welch_finish)
struct Pending {
    bool valid = false;
    const void *xd = nullptr;
    bool cplx = false;
    int64_t nsig = 0, nframes = 0, nmean = 0;
    int nfft = 0, hop = 0;
    Xf xf;
    const float *win_d = nullptr;
    const cf *Wf = nullptr;
    OnePass st;
    float *trend_f = nullptr;
    cf *cw = nullptr;
    double *sum_d = nullptr;
} g_pend;

struct ProfScope {   // brackets one kernel launch with HIP events on its stream (default: the launch stream) when profiling is on
    bool on;
    hipStream_t st;
    explicit ProfScope(hipStream_t s = nullptr) : on(g.profile), st(s ? s : g.stream) {
        if (on) (void)hipEventRecord(g.ev0, st);
    }
    ~ProfScope() {
        if (on) {
            (void)hipEventRecord(g.ev1, st);
            g.prof_valid = true;
        }
    }
};

int ensure_init() {
    if (g.ready) return 0;
    return sp_init(-1);
}

bool is_pow2(int64_t n) { return n >= 1 && (n & (n - 1)) == 0; }
int64_t next_pow2(int64_t n) {
    int64_t p = 1;
    while (p < n) p <<= 1;
    return p;
}

int get_twiddles(int64_t n, const cf **out) {
    auto it = g.twiddles.find(n);
    if (it != g.twiddles.end()) {
        *out = it->second;
        return 0;
    }
    std::vector<cf> h((size_t)n);
    for (int64_t m = 0; m < n; ++m) {
        const double a = -2.0 * M_PI * (double)m / (double)n;
        h[(size_t)m] = make_float2((float)cos(a), (float)sin(a));
    }
    cf *d = nullptr;
    HIPCHK(hipMalloc((void **)&d, sizeof(cf) * (size_t)n));
    HIPCHK(hipMemcpy(d, h.data(), sizeof(cf) * (size_t)n, hipMemcpyHostToDevice));
    g.twiddles[n] = d;
    *out = d;
    return 0;
}

// transform plan for length n handled inside one workgroup
int get_xf(int64_t n, Xf *xf) {
    if (n < 2) return fail("transform length %lld too short", (long long)n);
    if (is_pow2(n) && n <= SP_MAX_WG_FFT) {
        xf->L = (int)n;
        xf->blue = false;
        xf->tb.n = (int)n;
        xf->tb.chirp = xf->tb.bf = nullptr;
        return get_twiddles(n, &xf->tb.tw);
    }
    int64_t L = next_pow2(2 * n - 1);
    if (L < 16) L = 16;
    if (L > SP_MAX_WG_FFT)
        return fail("transform length %lld not supported yet (powers of two up to %d, other lengths up to %d)",
                    (long long)n, SP_MAX_WG_FFT, SP_MAX_WG_FFT / 2);
    xf->L = (int)L;
    xf->blue = true;
    xf->tb.n = (int)n;
    if (get_twiddles(L, &xf->tb.tw)) return -1;
    auto it = g.blue.find(n);
    if (it == g.blue.end()) {
        // chirp[m] = exp(-i pi m^2/n); bf = FFT_L(wrapped conj(chirp)) / L   (angles reduced mod 2n in integers)
        std::vector<cf> ch((size_t)n), bw((size_t)L, make_float2(0.f, 0.f));
        for (int64_t m = 0; m < n; ++m) {
            const int64_t q = (m * m) % (2 * n);
            const double a = M_PI * (double)q / (double)n;
            ch[(size_t)m] = make_float2((float)cos(a), (float)-sin(a));
            const cf b = make_float2((float)(cos(a) / (double)L), (float)(sin(a) / (double)L));
            bw[(size_t)m] = b;
            if (m > 0) bw[(size_t)(L - m)] = b;
        }
        BlueTab t;
        t.L = (int)L;
        HIPCHK(hipMalloc((void **)&t.chirp, sizeof(cf) * (size_t)n));
        HIPCHK(hipMalloc((void **)&t.bf, sizeof(cf) * (size_t)L));
        HIPCHK(hipMemcpy(t.chirp, ch.data(), sizeof(cf) * (size_t)n, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(t.bf, bw.data(), sizeof(cf) * (size_t)L, hipMemcpyHostToDevice));
        Xf p2;
        if (get_xf(L, &p2)) return -1;
        LAUNCHCHK(launch_fft_c2c(lc(), t.bf, t.bf, 1, 0, p2));
        HIPCHK(hipStreamSynchronize(g.stream));
        g.blue[n] = t;
        it = g.blue.find(n);
    }
    xf->tb.chirp = it->second.chirp;
    xf->tb.bf = it->second.bf;
    return 0;
}

// Small host tables (windows, filter spectra) live in a content-keyed device cache: a table is uploaded
// once into its own allocation and never overwritten, so asynchronous (mem=1) callers can reuse the
// host buffer immediately and repeated calls with the same window cost no copy and no synchronisation.
// (bookkeeping and eviction policy: table_cache.h.)  A miss on a full cache drops the least recently used entry that
// the current API call has not touched and that a pending sp_welch_accum does not hold; the device is synchronised
// before the entry is freed, because asynchronous (mem=1) work of an earlier call may still read it.
TableCache g_tables;

// every C-ABI entry point that touches device state holds this for its whole body: the library lock, and the start of
// a new "current call" for the table cache
struct ApiLock {
    std::lock_guard<std::mutex> lk;
    ApiLock() : lk(g.mu) { g_tables.begin_call(); }
};

// content hash of a host table (FNV-1a style, 8 bytes per step: a 4096-point window hashes in ~1 us)
uint64_t fnv1a(const void *p, size_t n, uint64_t h) {
    const unsigned char *b = (const unsigned char *)p;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, b + i, 8);
        h ^= w;
        h *= 1099511628211ull;
        h ^= h >> 29;
    }
    for (; i < n; ++i) {
        h ^= b[i];
        h *= 1099511628211ull;
    }
    return h;
}

// content hash of a WINDOW, memoised by comparing with a kept copy: a streamed Welch step asked for the same window's hash four
// times (window table, its spectrum, lobe test, COLA test: 3-5 us per pass over 16 KiB, half of the call's host time); a memcmp
// of 16 KiB is 0.2 us.  Called under the library lock.
uint64_t window_key(const float *win, int n) {
    struct Memo {
        std::vector<float> copy;
        uint64_t h = 0;
    };
    static Memo memo[4];
    static int next = 0;
    const size_t bytes = sizeof(float) * (size_t)n;
    for (Memo &m : memo)
        if ((int)m.copy.size() == n && memcmp(m.copy.data(), win, bytes) == 0) return m.h;
    Memo &m = memo[next];
    next = (next + 1) & 3;
    m.copy.assign(win, win + n);
    m.h = fnv1a(win, bytes, 1469598103934665603ull);
    return m.h;
}
static uint64_t mix_key(uint64_t h, uint64_t a, uint64_t b) {
    h ^= a * 0x9E3779B97F4A7C15ull;
    h *= 1099511628211ull;
    h ^= h >> 29;
    h ^= b;
    h *= 1099511628211ull;
    return h ^ (h >> 32);
}

void tables_release() {
    for (auto &kv : g_tables.map) (void)hipFree(kv.second.dev);
    g_tables.map.clear();
}

// device table keyed by the CONTENT of `key_data`.  upload != null: a miss uploads `bytes` from it; upload == null: a
// miss only allocates `bytes` and reports fresh = true (the caller fills the table, e.g. FFT(window) keyed by the window)
int get_table_keyed(uint64_t kind, const void *key_data, size_t key_bytes, const void *upload, size_t bytes, void **dev,
                    bool *fresh) {
    // (kinds 1 and 3 are keyed by a window's float32 values: the memoised hash)
    const uint64_t key = ((kind == 1 || kind == 3) && key_bytes % sizeof(float) == 0 && key_bytes > 0)
                             ? mix_key(window_key((const float *)key_data, (int)(key_bytes / sizeof(float))), kind, bytes)
                             : fnv1a(key_data, key_bytes, 1469598103934665603ull ^ (kind * 0x9E3779B97F4A7C15ull) ^ bytes);
    if (TableEntry *e = g_tables.find(key, bytes)) {
        *dev = e->dev;
        if (fresh) *fresh = false;
        return 0;
    }
    if (g_tables.full()) {
        const void *pinned[2] = {g_pend.valid ? (const void *)g_pend.win_d : nullptr, g_pend.valid ? (const void *)g_pend.Wf : nullptr};
        uint64_t victim;
        bool synced = false;
        while (g_tables.full() && g_tables.pick_victim(pinned, 2, &victim)) {
            if (!synced) HIPCHK(hipDeviceSynchronize());
            synced = true;
            (void)hipFree(g_tables.erase(victim));
        }   // nothing evictable: every entry is in use by this call -- grow past the cap rather than free live memory
    }
    void *d = nullptr;
    HIPCHK(hipMalloc(&d, bytes));
    if (upload) HIPCHK(hipMemcpy(d, upload, bytes, hipMemcpyHostToDevice));
    if (void *old = g_tables.insert(key, d, bytes)) {
        HIPCHK(hipDeviceSynchronize());
        (void)hipFree(old);
    }
    *dev = d;
    if (fresh) *fresh = true;
    return 0;
}
int get_table(uint64_t kind, const void *host, size_t bytes, void **dev, bool *fresh) {
    return get_table_keyed(kind, host, bytes, host, bytes, dev, fresh);
}

__global__ void k_set4(float *dst, float a, float b, float c, float d) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        dst[0] = a;
        dst[1] = b;
        dst[2] = c;
        dst[3] = d;
    }
}

// dst[c] = trend of src[c] re-based to start at sample `offset`: (m + s*offset, s)
__global__ void k_trend_shift(const float *src, float *dst, int count, double offset) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < count) {
        dst[4 * c + 0] = (float)((double)src[4 * c + 0] + (double)src[4 * c + 2] * offset);
        dst[4 * c + 1] = (float)((double)src[4 * c + 1] + (double)src[4 * c + 3] * offset);
        dst[4 * c + 2] = src[4 * c + 2];
        dst[4 * c + 3] = src[4 * c + 3];
    }
}

__global__ void k_xcorr_norm(const double *mom1, const double *mom2, int64_t n, double *out /*[4]*/) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double m1 = mom1[0], m2 = mom2[0];
        const double v1 = mom1[2] / (double)n - m1 * m1, v2 = mom2[2] / (double)n - m2 * m2;
        out[0] = m1;
        out[1] = m2;
        out[2] = 1.0 / ((double)n * sqrt(v1 > 0 ? v1 : 0) * sqrt(v2 > 0 ? v2 : 0));
        out[3] = 0;
    }
}

// device block: nsig trend records (4 floats each) followed by 8 doubles per signal
struct TrendBuf {
    float *f;
    double *d;
};
int get_trendbuf(int nsig, TrendBuf *tb) {
    const size_t fbytes = ((sizeof(float) * 4 * (size_t)nsig) + 63) & ~(size_t)63;
    if (g.trends.ensure(fbytes + sizeof(double) * 8 * (size_t)nsig + 64)) return -1;
    tb->f = (float *)g.trends.p;
    tb->d = (double *)((char *)g.trends.p + fbytes);
    return 0;
}
double *moments_scratch() {
    if (g.small.ensure(sizeof(double) * 8 * 4096 + 256)) return nullptr;
    return (double *)g.small.p;
}

// fill trend record `idx` for signal x: detrend 0 -> explicit (re,im), 1 -> mean, 2 -> LS line
int set_trend(const TrendBuf &tb, int idx, const void *x, bool cplx, int64_t n, int detrend, double re, double im) {
    if (detrend == 0) {
        hipLaunchKernelGGL(k_set4, dim3(1), dim3(64), 0, g.stream, tb.f + 4 * idx, (float)re, (float)im, 0.f, 0.f);
        HIPCHK(hipGetLastError());
        return 0;
    }
    double *scr = moments_scratch();
    if (!scr) return -1;
    LAUNCHCHK(launch_moments(lc(), x, cplx, n, detrend, scr, tb.d + 8 * idx, tb.f + 4 * idx));
    return 0;
}

int check_frames(const char *who, int64_t nsig, int nfft, int hop, int64_t nframes) {
    if (nfft < 2 || hop < 1 || nframes < 1) return fail("%s: bad nfft/hop/nframes", who);
    if ((nframes - 1) * (int64_t)hop + nfft > nsig)
        return fail("%s: %lld frames of %d with hop %d need %lld samples, signal has %lld", who, (long long)nframes, nfft,
                    hop, (long long)((nframes - 1) * (int64_t)hop + nfft), (long long)nsig);
    return 0;
}

int nbins_host(int n, int sided) {
    if (sided == SP_SIDED_ONE) return (n & 1) ? (n + 1) / 2 : n / 2;
    if (sided == SP_SIDED_HALF) return n / 2 + 1;
    return n;
}

bool env_flag(const char *name) {
    const char *v = getenv(name);
    return v && v[0] && v[0] != '0';
}


// ---- transforms longer than one workgroup ---------------------------------------------------------
bool wg_capable(int64_t n) {
    if (n < 2) return false;
    if (is_pow2(n)) return n <= SP_MAX_WG_FFT;
    int64_t L = next_pow2(2 * n - 1);
    return (L < 16 ? 16 : L) <= SP_MAX_WG_FFT;
}

int get_bigtw(int64_t N, BigTw *bt) {
    auto it = g.bigtw.find(N);
    if (it != g.bigtw.end()) {
        *bt = it->second;
        return 0;
    }
    int lg = 0;
    while (((int64_t)1 << lg) < N) ++lg;
    const int lb = lg < 13 ? lg : 13;
    const int64_t nlo = (int64_t)1 << lb, nhi = N >> lb;
    std::vector<cf> lo((size_t)nlo), hi((size_t)nhi);
    for (int64_t j = 0; j < nlo; ++j) {
        const double a = -2.0 * M_PI * (double)j / (double)N;
        lo[(size_t)j] = make_float2((float)cos(a), (float)sin(a));
    }
    for (int64_t j = 0; j < nhi; ++j) {
        const double a = -2.0 * M_PI * (double)j / (double)nhi;      // j * 2^lb / N
        hi[(size_t)j] = make_float2((float)cos(a), (float)sin(a));
    }
    cf *dlo = nullptr, *dhi = nullptr;
    HIPCHK(hipMalloc((void **)&dlo, sizeof(cf) * (size_t)nlo));
    HIPCHK(hipMalloc((void **)&dhi, sizeof(cf) * (size_t)nhi));
    HIPCHK(hipMemcpy(dlo, lo.data(), sizeof(cf) * (size_t)nlo, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dhi, hi.data(), sizeof(cf) * (size_t)nhi, hipMemcpyHostToDevice));
    bt->hi = dhi;
    bt->lo = dlo;
    bt->lb = lb;
    bt->mod = 0;
    g.bigtw[N] = *bt;
    return 0;
}

// N = 2^k, 2^14 <= N <= 2^26:  x viewed as [N1][N2]; transpose, N2 row FFTs of N1 (+ twiddle W_N^{n2 k1}),
// transpose, N1 row FFTs of N2, transpose back to natural order.  5 passes over the data (80 B / point).
// three-pass form available for N (then hmask may be fused into the first pass of an inverse transform)
bool big_three_pass(int64_t N) {
    return is_pow2(N) && N >= ((int64_t)1 << 20) && !env_flag("SP_BIGFFT_5PASS") && !env_flag("SP_BIGFFT_2PASS");
}

// fused first-pass input / last-pass output of ONE three-pass transform (see ColsIn / RowsOut in kernels.h)
struct BigFuse {
    ColsIn ci = ColsIn{0, nullptr, nullptr, nullptr, 0};
    RowsOut ro = RowsOut{nullptr, 0, 0, nullptr};
};
int dev_fft_big_pow2(const cf *in, cf *out, int64_t N, int inverse, int hmask = 0, int64_t batch = 1, const BigFuse *fz = nullptr) {
    int lg = 0;
    while (((int64_t)1 << lg) < N) ++lg;
    if (((int64_t)1 << lg) != N || lg > SP_MAX_BIG_LOG2) return fail("internal: dev_fft_big_pow2(%lld)", (long long)N);
    if (batch < 1) return fail("internal: dev_fft_big_pow2 batch");
    const int64_t N1 = (int64_t)1 << ((lg + 1) / 2), N2 = N / N1;
    Xf x1, x2;
    if (get_xf(N1, &x1) || get_xf(N2, &x2)) return -1;
    BigTw bt;
    if (get_bigtw(N, &bt)) return -1;
    const float sc = inverse ? (float)(1.0 / (double)N) : 1.f;
    if ((hmask || fz) && !big_three_pass(N)) return fail("internal: fused Hilbert mask / pack need the three-pass transform");
    if (fz && batch != 1) return fail("internal: fused long transform is a single transform");
    if (big_three_pass(N)) {
        // three passes over whole lines, N = A B C (kernels.h: k_fft_cols / k_fft_rows_rev); a batch runs row by row
        // (each pass already fills the chip at these lengths)
        if (g.bigT.ensure(sizeof(cf) * (size_t)N)) return -1;
        cf *tmp = (cf *)g.bigT.p;
        // the two COLUMN passes want short transforms: a workgroup then owns many adjacent columns and every access is a
        // whole 128-byte line or more (256 points: 16 columns; 512 points would be 8 columns = 64-byte segments, measured
        // 1.4 TB/s for that pass at 2^25 points) -- A, B <= 256, the rest goes to the contiguous row pass
        int la_ = (lg + 2) / 3;
        if (la_ > 8) la_ = 8;
        int lb_ = (lg - la_ + 1) / 2;
        if (lb_ > 8) lb_ = 8;
        const int lc_ = lg - la_ - lb_;
        const int64_t A = (int64_t)1 << la_, B = (int64_t)1 << lb_, C = (int64_t)1 << lc_;
        Xf xa, xb, xc;
        if (get_xf(A, &xa) || get_xf(B, &xb) || get_xf(C, &xc)) return -1;
        for (int64_t b = 0; b < batch; ++b) {
            if (fz) {
                LAUNCHCHK(launch_fft_cols(lc(), in, tmp, B * C, 1, B * C, 0, 1, inverse, xa, bt, hmask ? N : 0, fz->ci));
                LAUNCHCHK(launch_fft_cols(lc(), tmp, tmp, C, A, C, B * C, A, 0, xb, bt));
                LAUNCHCHK(launch_fft_rows_rev(lc(), tmp, out, A, B, inverse, sc, xc, fz->ro));
                continue;
            }
            LAUNCHCHK(launch_fft_cols(lc(), in + b * N, tmp, B * C, 1, B * C, 0, 1, inverse, xa, bt, hmask ? N : 0));
            LAUNCHCHK(launch_fft_cols(lc(), tmp, tmp, C, A, C, B * C, A, 0, xb, bt));
            LAUNCHCHK(launch_fft_rows_rev(lc(), tmp, out + b * N, A, B, inverse, sc, xc));
        }
        return 0;
    }
    if (env_flag("SP_BIGFFT_2PASS") && N1 >= 16 && N2 >= 16) {
        // experiment, off by default (measured 0.90-1.19 ms against 0.83 for a 2^24-point Hilbert: the column reads run at
        // 1.3 TB/s even with XCD-local adjacent columns).  Two strided passes, no explicit transposes (x viewed as
        // [N1][N2], n = n1 N2 + n2):
        //   A: for every column n2: FFT over n1, times W_N^{n2 k1}  -> tmp[n2][k1]            (column read, row write)
        //   B: for every column k1 of tmp: FFT over n2               -> out[k1 + N1 k2]       (column read, column write)
        if (g.bigT.ensure(sizeof(cf) * (size_t)N)) return -1;
        cf *tmp = (cf *)g.bigT.p;
        for (int64_t b = 0; b < batch; ++b) {
            LAUNCHCHK(launch_fft_strided(lc(), in + b * N, tmp, N2, 1, N2, N1, 1, inverse, 0, 1.f, x1, bt));
            LAUNCHCHK(launch_fft_strided(lc(), tmp, out + b * N, N1, 1, N1, 1, N1, 0, inverse, sc, x2, BigTw{nullptr, nullptr, 0, 0}));
        }
        return 0;
    }
    // N = 2^k, 2^14 <= N < 2^20:  x viewed as [N1][N2]; transpose, N2 row FFTs of N1 (+ twiddle W_N^{n2 k1}), transpose,
    // N1 row FFTs of N2, transpose back to natural order: 5 passes over the data (80 B / point).  The whole batch of
    // transforms goes through each of the five launches together (rows numbered modulo N2 for the twiddle).
    if (batch > 65535) return fail("internal: dev_fft_big_pow2 batch %lld", (long long)batch);
    if (g.bigT.ensure(sizeof(cf) * (size_t)N * (size_t)batch)) return -1;
    cf *tmp = (cf *)g.bigT.p;
    BigTw btb = bt;
    btb.mod = N2;
    if (in != out) {
        LAUNCHCHK(launch_transpose_c(lc(), in, out, N1, N2, inverse, 1.f, batch));        // out[n2][n1]
        LAUNCHCHK(launch_fft_c2c(lc(), out, out, N2 * batch, 0, x1, btb));                 // A[n2][k1] W^{n2 k1}
        LAUNCHCHK(launch_transpose_c(lc(), out, tmp, N2, N1, 0, 1.f, batch));              // tmp[k1][n2]
        LAUNCHCHK(launch_fft_c2c(lc(), tmp, tmp, N1 * batch, 0, x2));                      // B[k1][k2]
        LAUNCHCHK(launch_transpose_c(lc(), tmp, out, N1, N2, inverse, sc, batch));         // out[k2][k1]
    } else {
        if (g.blueB.ensure(sizeof(cf) * (size_t)N * (size_t)batch)) return -1;     // second temporary for the in-place form
        cf *t2 = (cf *)g.blueB.p;
        LAUNCHCHK(launch_transpose_c(lc(), in, tmp, N1, N2, inverse, 1.f, batch));
        LAUNCHCHK(launch_fft_c2c(lc(), tmp, tmp, N2 * batch, 0, x1, btb));
        LAUNCHCHK(launch_transpose_c(lc(), tmp, t2, N2, N1, 0, 1.f, batch));
        LAUNCHCHK(launch_fft_c2c(lc(), t2, t2, N1 * batch, 0, x2));
        LAUNCHCHK(launch_transpose_c(lc(), t2, out, N1, N2, inverse, sc, batch));
    }
    return 0;
}

int get_blue_big(int64_t n, BlueTab *t) {
    auto it = g.blue_big.find(n);
    if (it != g.blue_big.end()) {
        *t = it->second;
        return 0;
    }
    const int64_t L = next_pow2(2 * n - 1);
    std::vector<cf> ch((size_t)n), bw((size_t)L, make_float2(0.f, 0.f));
    for (int64_t m = 0; m < n; ++m) {
        const int64_t q = (int64_t)(((__int128)m * m) % (2 * n));
        const double a = M_PI * (double)q / (double)n;
        const float c = (float)cos(a), sn = (float)sin(a);
        ch[(size_t)m] = make_float2(c, -sn);
        bw[(size_t)m] = make_float2(c, sn);
        if (m > 0) bw[(size_t)(L - m)] = make_float2(c, sn);
    }
    BlueTab nt;
    nt.L = 0;
    HIPCHK(hipMalloc((void **)&nt.chirp, sizeof(cf) * (size_t)n));
    HIPCHK(hipMalloc((void **)&nt.bf, sizeof(cf) * (size_t)L));
    HIPCHK(hipMemcpy(nt.chirp, ch.data(), sizeof(cf) * (size_t)n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(nt.bf, bw.data(), sizeof(cf) * (size_t)L, hipMemcpyHostToDevice));
    if (dev_fft_big_pow2(nt.bf, nt.bf, L, 0)) return -1;
    HIPCHK(hipStreamSynchronize(g.stream));
    g.blue_big[n] = nt;
    *t = nt;
    return 0;
}

// any length, device pointers.  in == out allowed.  Batches of long transforms go through the multi-kernel paths in
// slices of at most SP_LONG_SLICE_BYTES of work buffer (every launch then covers a whole slice).
#define SP_LONG_SLICE_BYTES ((size_t)256 << 20)
int dev_fft_any(const cf *in, cf *out, int64_t n, int64_t batch, int inverse) {
    if (n == 1) {
        if (in != out) HIPCHK(hipMemcpyAsync(out, in, sizeof(cf) * (size_t)batch, hipMemcpyDeviceToDevice, g.stream));
        return 0;
    }
    if (wg_capable(n)) {
        Xf xf;
        if (get_xf(n, &xf)) return -1;
        LAUNCHCHK(launch_fft_c2c(lc(), in, out, batch, inverse, xf));
        return 0;
    }
    const int64_t L = is_pow2(n) ? n : next_pow2(2 * n - 1);
    if (L > ((int64_t)1 << SP_MAX_BIG_LOG2))
        return fail("transform length %lld needs a %lld-point multi-pass transform; the limit is 2^%d", (long long)n,
                    (long long)L, SP_MAX_BIG_LOG2);
    int64_t slice = (int64_t)(SP_LONG_SLICE_BYTES / (sizeof(cf) * (size_t)L));
    if (slice < 1) slice = 1;
    if (slice > 32768) slice = 32768;
    if (is_pow2(n)) {
        for (int64_t b0 = 0; b0 < batch; b0 += slice) {
            const int64_t m = batch - b0 < slice ? batch - b0 : slice;
            if (dev_fft_big_pow2(in + b0 * n, out + b0 * n, n, inverse, 0, m)) return -1;
        }
        return 0;
    }
    BlueTab bt;
    if (get_blue_big(n, &bt)) return -1;
    if (slice > batch) slice = batch;
    if (g.blueA.ensure(sizeof(cf) * (size_t)L * (size_t)slice)) return -1;
    cf *A = (cf *)g.blueA.p;
    for (int64_t b0 = 0; b0 < batch; b0 += slice) {
        const int64_t m = batch - b0 < slice ? batch - b0 : slice;
        LAUNCHCHK(launch_blue_pre(lc(), in + b0 * n, bt.chirp, n, L, inverse, A, m));
        if (dev_fft_big_pow2(A, A, L, 0, 0, m)) return -1;
        LAUNCHCHK(launch_cmul_vec(lc(), A, bt.bf, L, 0, A, m));
        if (dev_fft_big_pow2(A, A, L, 1, 0, m)) return -1;
        LAUNCHCHK(launch_blue_post(lc(), A, bt.chirp, n, inverse, inverse ? (float)(1.0 / (double)n) : 1.f, out + b0 * n, m, L));
    }
    return 0;
}


// ---- segments longer than one workgroup transform (k_long.hip) -----------------------------------
// frames per chunk: the chunk's spectra (m x nfft complex64) stay below SP_LONG_CHUNK_BYTES
#define SP_LONG_CHUNK_BYTES ((size_t)192 << 20)
int64_t long_chunk_frames(int nfft, int64_t nframes) {
    int64_t m = (int64_t)(SP_LONG_CHUNK_BYTES / (sizeof(cf) * (size_t)nfft));
    if (m > 32768) m = 32768;
    if (m > nframes) m = nframes;
    if (m < 1) m = 1;
    return m;
}
// spectra of frames [f0, f0+m) of one signal -> buf[m][nfft] (natural bin order, unscaled); segmean: 1 / 2 = every
// frame's own mean / least-squares line is removed first; pseg_d (optional, zeroed): per-frame time-domain power
int long_spectra(const void *xd, bool cplx, const float *win_d, int nfft, int hop, int64_t f0, int64_t m, const float *trend,
                 bool lin, int segmean, cf *buf, double *pseg_d) {
    const float *rec = nullptr;
    if (segmean) {
        if (g.longrec.ensure(sizeof(float) * 4 * (size_t)m)) return -1;
        LAUNCHCHK(launch_long_segstats(lc(), xd, cplx, f0, m, hop, nfft, segmean, (float *)g.longrec.p));
        rec = (const float *)g.longrec.p;
    }
    LAUNCHCHK(launch_long_pack(lc(), xd, cplx, win_d, nfft, hop, f0, m, trend, lin, rec, buf, pseg_d));
    return dev_fft_any(buf, buf, nfft, m, 0);
}


// ---- one-pass Welch: accumulate, then finish with a (possibly global) mean ------------------------
// Wf = FFT(window) on the device, cached under the window's content
int get_window_spectrum(const float *win, int nfft, const Xf &xf, void **Wf_d) {
    bool fresh = false;
    if (get_table_keyed(3, win, sizeof(float) * (size_t)nfft, nullptr, sizeof(cf) * (size_t)nfft, Wf_d, &fresh)) return -1;
    if (fresh) {
        std::vector<cf> wc((size_t)nfft);
        for (int i = 0; i < nfft; ++i) wc[(size_t)i] = make_float2(win[i], 0.f);
        HIPCHK(hipMemcpy(*Wf_d, wc.data(), sizeof(cf) * (size_t)nfft, hipMemcpyHostToDevice));
        LAUNCHCHK(launch_fft_c2c(lc(), (const cf *)*Wf_d, (cf *)*Wf_d, 1, 0, xf));
    }
    return 0;
}

// the nfft-4096 Welch shapes run as a pipeline of specialised waves (k_welch_pipe.hip): one 768-thread workgroup per CU.
// The pipeline spends 4 periods per workgroup filling and draining; the symmetric kernel takes ~4 periods per frame and workgroup at
// two workgroups per CU, so the pipeline wins from 2 frames per CU on (round 3, tools/minfpc_ab.sh, step / kernel ms: 2^24 samples
// 0.054 / 0.045 against 0.071 / 0.060, 2^22: 0.045 / 0.024 against 0.062 / 0.036, 2^20: 0.038 / 0.019 against 0.053 / 0.020; the
// threshold had been 32 per CU, which sent every shard below 2^25 samples to the symmetric kernel).  SP_PIPE_MINFPC sets the
// threshold; SP_WELCH_PIPE=0 turns the pipeline off, 2 forces it for any frame count (the tests' way to reach its tail handling).
#ifndef SP_PIPE_MINFPC_DEFAULT
#define SP_PIPE_MINFPC_DEFAULT 2
#endif
static int welch_pipe_gpc() {
    static const int v = [] {
        const char *e = getenv("SP_PIPE_GPC");
        const int k = e ? atoi(e) : 0;
        return k > 0 ? k : 1;
    }();
    return v;
}
static int welch_pipe_mode() {
    static const int mode = [] {
        const char *e = getenv("SP_WELCH_PIPE");
        return e ? atoi(e) : SP_WELCH_PIPE_DEFAULT;
    }();
    return mode;
}
static bool welch_pipe_wanted(const Xf &xf, int hop, int64_t nframes) {
    const int mode = welch_pipe_mode();
    static const int64_t minfpc = getenv("SP_PIPE_MINFPC") ? atoll(getenv("SP_PIPE_MINFPC")) : SP_PIPE_MINFPC_DEFAULT;   // frames per CU
    return welch_pipe_eligible(xf, hop) && (mode >= 2 || (mode == 1 && nframes >= minfpc * (int64_t)g.ncu - minfpc));
}

// FFT(window) on the host in float64, to find out whether it is confined to the bins ks = -K .. K, K <= 3 (periodic cosine-sum
// windows: Hann, Hamming, Blackman, Nuttall ...): everything outside below 1e-9 of the peak.  Cached under the window's content.
static bool cog_window_lobe(const float *win, int n, CogLobe *lb) {
    struct Entry {
        uint64_t key;
        int n;
        bool ok;
        CogLobe lb;
    };
    static std::vector<Entry> cache;
    const uint64_t h = mix_key(window_key(win, n), 11, 0);
    for (const Entry &e : cache)
        if (e.key == h && e.n == n) {
            *lb = e.lb;
            return e.ok;
        }
    Entry e{h, n, false, CogLobe{}};
    // the lobe bins exactly, and the energy outside them by Parseval: sum |W|^2 = n sum w^2
    double tot = 0.0;
    for (int i = 0; i < n; ++i) tot += (double)win[i] * (double)win[i];
    tot *= (double)n;
    double in = 0.0, peak = 0.0;
    CogLobe l{};
    l.K = 3;
    for (int ks = -3; ks <= 3; ++ks) {
        double re = 0.0, im = 0.0;
        for (int i = 0; i < n; ++i) {
            const double a = -2.0 * M_PI * (double)(((int64_t)ks * i) % n) / (double)n;
            re += (double)win[i] * std::cos(a);
            im += (double)win[i] * std::sin(a);
        }
        l.wr[ks + 3] = re;
        l.wi[ks + 3] = im;
        in += re * re + im * im;
        if (re * re + im * im > peak) peak = re * re + im * im;
    }
    e.ok = peak > 0.0 && std::fabs(tot - in) <= 1e-9 * tot;      // (float32 window values: the tables' own rounding leaks ~1e-14)
    e.lb = l;
    if (cache.size() < 64) cache.push_back(e);
    *lb = l;
    return e.ok;
}


// does the window add up to a constant at this hop (COLA)?  c = sum_j w[n + j hop] for every n < hop, host float64 on the float32
// table (periodic Hann / Hamming at 50 % and 75 %: yes; Blackman at 50 %: no).  Cached under the window's content and the hop.
static bool window_cola(const float *win, int n, int hop, double *c_out) {
    struct Entry {
        uint64_t key;
        int n, hop;
        bool ok;
        double c;
    };
    static std::vector<Entry> cache;
    if (hop < 1 || n % hop != 0) return false;
    const uint64_t h = mix_key(window_key(win, n), 7, (uint64_t)hop);
    for (const Entry &e : cache)
        if (e.key == h && e.n == n && e.hop == hop) {
            *c_out = e.c;
            return e.ok;
        }
    double lo = 1e300, hi = -1e300, sum = 0.0;
    for (int i = 0; i < hop; ++i) {
        double c = 0.0;
        for (int j = i; j < n; j += hop) c += (double)win[j];
        lo = c < lo ? c : lo;
        hi = c > hi ? c : hi;
        sum += c;
    }
    Entry e{h, n, hop, false, sum / (double)hop};
    e.ok = e.c > 0.0 && (hi - lo) <= 2e-6 * e.c;          // (float32 table values: their own rounding leaves ~1e-7)
    if (cache.size() < 64) cache.push_back(e);
    *c_out = e.c;
    return e.ok;
}

// request for the epilogue in the same call (k_op_fused: column sums + finish / export in ONE launch; taken when the window's
// spectrum is confined to the bins -3 .. 3 -- every cosine-sum window -- unless SP_OP_UNFUSED=1): done = true on return when
// `out` has been produced, otherwise the caller runs k_op_finish as before
struct FusedOut {
    bool export_state;
    int sided;
    double scale;          // already divided by the frame count
    double *out;
    bool done;
    // pipelined sharded PSD: the previous step's summed state, applied by the same launch behind `prev_wait` (its all-reduce)
    OpPrev prev = OpPrev{nullptr, nullptr, nullptr, 0, 0.0};
    hipEvent_t prev_wait = nullptr;
    bool prev_done = false;
};
unsigned *get_ticket(Scratch &t) {
    if (!t.p) {                         // 9 counters, 64 bytes apart (k_op_fused)
        if (t.ensure(1024)) return nullptr;
        if (hipMemset(t.p, 0, 1024) != hipSuccess) return nullptr;
    }
    return (unsigned *)t.p;
}
// streaming engine (sp_welch_dist_*): a scratch set of its own per step parity, and the epilogue launched on another stream
// behind an event recorded after the main kernel -- so that it runs beside the NEXT step's main kernel
// main / ev_in (optional): the main kernel goes to a lane of the engine's own instead of the launch stream, ordered behind an event
// recorded on the launch stream just before it -- so that the NEXT step's main kernel (other lane) backfills the CUs as this
// one's workgroups retire instead of waiting behind its last one
struct SplitLaunch {
    Scratch *work, *onepass, *trend, *ticket;
    hipStream_t epi;
    hipEvent_t ev_main;
    hipStream_t main = nullptr;
    hipEvent_t ev_in = nullptr;
    int reserve_cus = 0;       // CUs left to the collective's kernel: the main kernel is partitioned over ncu - reserve_cus
};

// want_sum: also produce the shard's plain sample sum (split ABI, one more tiny launch); without it the finish kernel
// derives the shard mean itself
int welch_accum_locked(const void *xd, bool cplx, int64_t nsig, const float *win, int nfft, int hop, int64_t nframes,
                       int64_t nmean, bool want_sum, FusedOut *fo = nullptr, const SplitLaunch *sl = nullptr) {
    Scratch &S_work = sl ? *sl->work : g.work, &S_one = sl ? *sl->onepass : g.onepass, &S_trend = sl ? *sl->trend : g.pend_trend;
    Scratch &S_ticket = sl ? *sl->ticket : g.ticket;
    if (!wg_capable(nfft))
        return fail("sharded / split Welch PSD (sp_welch_accum, sp_welch_export, sp_welch_dist_*): segments longer than one workgroup "
                    "transform are not sharded (nfft = %d; powers of two in [256, %d] with hop = nfft/4, nfft/2 or nfft) -- the "
                    "long-segment regime has few, large frames: run sp_welch_psd on one GPU", nfft, SP_MAX_WG_FFT);
    Xf xf;
    if (get_xf(nfft, &xf)) return -1;
    if (!welch_carry_eligible(xf, hop, false))
        return fail("one-pass Welch needs a power-of-two nfft in [256, %d] and hop = nfft/4, nfft/2 or nfft (got nfft=%d hop=%d)",
                    SP_MAX_WG_FFT, nfft, hop);
    if (nmean < 1 || nmean > nsig) return fail("sp_welch_accum: nmean must be in [1, nsig]");
    void *win_d;
    if (get_table(1, win, sizeof(float) * (size_t)nfft, &win_d, nullptr)) return -1;
    void *Wf_d;
    if (get_window_spectrum(win, nfft, xf, &Wf_d)) return -1;
    // its own trend record: calls between sp_welch_accum and sp_welch_finish reuse the shared one
    if (S_trend.ensure(256)) return -1;
    TrendBuf tb{(float *)S_trend.p, nullptr};
    // real input at hop = nfft/2: two frames per transform on the pipeline (k_welch_pipe modes 3/4), partitioned over frame PAIRS
    const bool realpair = !cplx && 2 * hop == nfft && nframes >= 2 && !env_flag("SP_NO_REALPAIR") &&
                          welch_pipe_wanted(xf, hop, (nframes + 1) / 2);
    // (not at hop = nfft: the one-pass front role with 16 new samples per thread and their block sums spills 20 registers and
    //  runs at half the symmetric kernel's rate, 1.00 against 0.50 ms at 2^28 samples; the plain mode is faster there, 0.40 / 0.44)
    //  (SP_WELCH_PIPE=2 still forces it: tests/test_gpu_pipe.py keeps the instantiation correct)
    const bool pipe = realpair || ((hop != nfft || welch_pipe_mode() >= 2) && welch_pipe_wanted(xf, hop, nframes));
    const int ncu_p = (sl && sl->reserve_cus > 0 && sl->reserve_cus < g.ncu) ? g.ncu - sl->reserve_cus : g.ncu;   // (pipeline: one workgroup per CU)
    const RunPart rp = realpair ? run_partition(xf.L, (nframes + 1) / 2, ncu_p, welch_pipe_gpc())
                                : (pipe ? run_partition(xf.L, nframes, ncu_p, welch_pipe_gpc()) : run_partition(xf.L, nframes, g.ncu));
    if (S_work.ensure(sizeof(float) * (size_t)rp.groups * xf.L)) return -1;
    const size_t sp_bytes = sizeof(cf) * (size_t)rp.groups * (size_t)hop;
    const size_t st_doubles = (size_t)nfft + 2 * (size_t)hop + 8;
    const size_t sp_pad = (sp_bytes + 255) & ~(size_t)255;
    const size_t st_pad = (sizeof(double) * st_doubles + 255) & ~(size_t)255;
    if (S_one.ensure(sp_pad + st_pad + sizeof(cf) * (size_t)nfft)) return -1;
    cf *spartial = (cf *)S_one.p;
    double *stp = (double *)((char *)S_one.p + sp_pad);
    OnePass st;
    st.A = stp;
    st.Sl = stp + nfft;
    st.tot = st.Sl + 2 * (size_t)hop;
    st.dlt = st.tot + 2;
    double *sum_d = st.dlt + 2;
    cf *cw = (cf *)((char *)S_one.p + sp_pad + st_pad);
    float *partial = (float *)S_work.p;
    double *est = moments_scratch();
    if (!est) return -1;
#if !SP_EST_IN_KERNEL
    if (!pipe) LAUNCHCHK(launch_op_estimate(lc(), xd, cplx, nsig, est, tb.f));
#else
    (void)est;                       // the main kernel estimates mu0 itself and publishes it in tb.f
#endif
    st.sym = realpair ? 1 : 0;
    // mode 9 of the pipeline kernel: the window's lobe bins of sum_g X_g accumulated by the BACK role (4 VALU per frame) instead of
    // the front role's block sums (16), for cosine-sum windows that are COLA at this hop and the one-launch epilogue (SP_OP_NOLOBESUM=1
    // keeps the block sums)
    CogLobe lobe{};
    double cola_c = 0.0;
    const bool fused_ok = fo && !want_sum && !env_flag("SP_OP_UNFUSED") && cog_window_lobe(win, nfft, &lobe);
    const bool lobesum = fused_ok && pipe && !realpair && hop != nfft && !env_flag("SP_OP_NOLOBESUM") && window_cola(win, nfft, hop, &cola_c);
    if (!lobesum) cola_c = 0.0;
    // the main kernel's stream: the launch stream, or the engine's lane behind everything enqueued on the launch stream so far
    // (the caller's producer of x, the table uploads above, the wait for the epilogue that last used this scratch set)
    const LaunchCtx mc = (sl && sl->main) ? LaunchCtx{sl->main, g.ncu} : lc();
    if (sl && sl->main) {
        HIPCHK(hipEventRecord(sl->ev_in, g.stream));
        HIPCHK(hipStreamWaitEvent(sl->main, sl->ev_in, 0));
    }
    // (streaming engine: the main kernel's event as the launch's own completion signal, not a record behind it)
    const bool stop_ev = sl && pipe && !g.profile && !env_flag("SP_DIST_RECORD_EVENT");
    if (pipe) {
        ProfScope ps(mc.stream);
        LaunchCtx mcs = mc;
        if (stop_ev) mcs.stop = sl->ev_main;
        LAUNCHCHK(launch_welch_pipe(mcs, xd, cplx, (const float *)win_d, hop, nframes, tb.f, xf, partial, rp, spartial,
                                    realpair ? 3 : (lobesum ? 9 : 0)));
        g.last_kernel = realpair ? "k_welch_pipe(onepass,realpair)" : (lobesum ? "k_welch_pipe(onepass,lobesum)" : "k_welch_pipe(onepass)");
    } else {
        ProfScope ps(mc.stream);
        LAUNCHCHK(launch_welch(mc, xd, cplx, (const float *)win_d, hop, nframes, tb.f, false, xf, partial, rp, true,
                               spartial, &g.last_kernel));
    }
    if (!want_sum) st.dlt = nullptr;
    // the epilogue's launch context: the launch stream, or the engine's epilogue stream behind the main kernel's event
    LaunchCtx ec = lc();
    if (sl) {
        if (!stop_ev) HIPCHK(hipEventRecord(sl->ev_main, mc.stream));
        HIPCHK(hipStreamWaitEvent(sl->epi, sl->ev_main, 0));
        ec = LaunchCtx{sl->epi, g.ncu};
    }
    if (fused_ok) {
        unsigned *ticket = get_ticket(S_ticket);
        if (!ticket) return fail("ticket allocation failed");
        if (fo->prev.st && fo->prev_wait) HIPCHK(hipStreamWaitEvent(ec.stream, fo->prev_wait, 0));
        LAUNCHCHK(launch_op_fused(ec, xd, cplx, tb.f, (const float *)win_d, partial, spartial, rp.groups, nfft, hop, nframes, nmean,
                                  st, ticket, lobe, nullptr, fo->sided, fo->scale, fo->out, fo->export_state, fo->prev,
                                  sl != nullptr && !env_flag("SP_OPF_HEAVY"), cola_c));
        fo->done = true;
        fo->prev_done = fo->prev.st != nullptr;
        g_pend.valid = false;
        return 0;
    }
    LAUNCHCHK(launch_op_reduce(ec, xd, cplx, tb.f, partial, spartial, rp.groups, xf, hop, nframes, nmean, st,
                               want_sum ? sum_d : nullptr));
    g_pend.valid = true;
    g_pend.xd = xd;
    g_pend.cplx = cplx;
    g_pend.nsig = nsig;
    g_pend.nframes = nframes;
    g_pend.nmean = nmean;
    g_pend.nfft = nfft;
    g_pend.hop = hop;
    g_pend.xf = xf;
    g_pend.win_d = (const float *)win_d;
    g_pend.Wf = (const cf *)Wf_d;
    g_pend.st = st;
    g_pend.trend_f = tb.f;
    g_pend.cw = cw;
    g_pend.sum_d = sum_d;
    return 0;
}

int welch_finish_locked(const double *mean_d /*device or null*/, int64_t frames_total, int sided, double scale,
                        double *out_d, const LaunchCtx *ctx = nullptr) {
    if (!g_pend.valid) return fail("sp_welch_finish: no pending sp_welch_accum");
    g_pend.valid = false;
    LAUNCHCHK(launch_op_finish(ctx ? *ctx : lc(), g_pend.xd, g_pend.cplx, g_pend.trend_f, g_pend.win_d, g_pend.st, mean_d, g_pend.nmean,
                               g_pend.xf, g_pend.hop, g_pend.nframes, g_pend.cw, g_pend.Wf, sided,
                               scale / (double)frames_total, out_d));
    return 0;
}

// ---- RCCL inside the library: one process per GPU, the library owns a communicator and a stream for the collective ---------
// The functions come from the RCCL that is already in the process when there is one (torch ships its own copy under the
// same soname, so dlopen returns that instance), else from /opt/rocm/lib.
struct Rccl {
    void *dl = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitRankConfig)(ncclComm_t *, int, ncclUniqueId, int, void *) = nullptr;      // optional (NCCL >= 2.17)
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
} rccl;

int rccl_load() {
    if (rccl.dl) return 0;
    const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    void *h = nullptr;
    for (const char *nm : names)
        if ((h = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return fail("RCCL not found (librccl.so.1): %s", dlerror());
    *(void **)&rccl.GetUniqueId = dlsym(h, "ncclGetUniqueId");
    *(void **)&rccl.CommInitRank = dlsym(h, "ncclCommInitRank");
    *(void **)&rccl.CommInitRankConfig = dlsym(h, "ncclCommInitRankConfig");
    *(void **)&rccl.CommDestroy = dlsym(h, "ncclCommDestroy");
    *(void **)&rccl.AllReduce = dlsym(h, "ncclAllReduce");
    *(void **)&rccl.GetErrorString = dlsym(h, "ncclGetErrorString");
    if (!rccl.GetUniqueId || !rccl.CommInitRank || !rccl.CommDestroy || !rccl.AllReduce || !rccl.GetErrorString)
        return fail("librccl lacks an expected symbol");
    rccl.dl = h;
    return 0;
}
#define NCCLCHK(expr)                                                                                 \
    do {                                                                                              \
        ncclResult_t r_ = (expr);                                                                     \
        if (r_ != ncclSuccess) return fail("%s failed: %s (%s:%d)", #expr, rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

struct Comm {
    ncclComm_t comm = nullptr;
    int world = 0, rank = -1;
    int ctas = 0;              // workgroups the communicator's kernels were limited to (0: RCCL's own choice)
} gcomm;

// RCCL's collective kernel cannot run beside k_welch_pipe: rcclGenericKernel takes 261-280 VGPRs + 17-32 AGPRs per lane at 256
// threads (code object metadata of librccl for gfx950) against the 80 per SIMD the pipeline leaves -- its workgroups would wait
// for pipeline workgroups to retire and then hold back the NEXT main kernel's on those CUs for the collective's duration (all 256
// workgroups of a main kernel are equal, so its makespan grows by that much).  Hence, with a communicator of more than one rank:
// the communicator is created with at most SP_DIST_RCCL_CTAS workgroups (default 4: the state is 160 KiB, latency-bound) and the
// streaming engine partitions its main kernels over ncu - that many CUs, so the collective always finds CUs of its own and
// runs beside the main kernel like the light epilogue does.  SP_DIST_RESERVE_CUS overrides the number of CUs left free (also for
// one rank, which is how the one-GPU tests reach the path), 0 turns it off.
static int dist_rccl_ctas() {
    const char *e = getenv("SP_DIST_RCCL_CTAS");
    const int v = e ? atoi(e) : 4;
    return v < 0 ? 0 : (v > 64 ? 64 : v);
}
static int dist_reserved_cus() {
    if (const char *e = getenv("SP_DIST_RESERVE_CUS")) {
        const int v = atoi(e);
        return v < 0 ? 0 : (v > g.ncu / 2 ? g.ncu / 2 : v);
    }
    // (twice the workgroup limit, at least 8: a limit RCCL rounds up must still find room; 8 CUs cost 1.2-1.5 % of the main
    //  kernel, tools/reserve_ab.sh: 0.5495 -> 0.5550 / 0.5580 ms at 2^28 samples for 4 / 8, 0.0819 -> 0.0825 / 0.0829 at 2^25)
    //  A multiple of 8 = one CU on every XCD: workgroups are dealt to the XCDs in turn, a reserve of 4 leaves XCDs 0-3 full --
    //  exactly where the collective's first workgroups go (tools/ubench/coexec_rccl.hip))
    const int want = 2 * gcomm.ctas > 8 ? 2 * gcomm.ctas : 8;
    return (gcomm.comm && gcomm.world > 1) ? (want + 7) / 8 * 8 : 0;
}

// ---- the streaming engine behind sp_welch_dist_submit / _flush ------------------------------------------------------------
// Step k: the main kernel (k_welch_pipe / k_welch_carry) on the launch stream A into the scratch set of parity k & 1; an event;
// the epilogue on the engine's stream B behind it: k_op_fused (with a communicator: EXPORT of this step's state + the apply of
// step k-1's all-reduced state, then ncclAllReduce of this step's state, all B-ordered).  A's next main kernel does not depend
// on B, so the epilogue (15-20 us of mostly one workgroup) and the collective run BESIDE it (tools/ubench/coexec.hip: two
// kernels from two streams share the CUs when the first leaves registers and wave slots free; the main kernel takes 408 of a
// SIMD's 512 VGPRs and 12 of a CU's 32 wave slots).  Only then does A wait for the PREVIOUS step's epilogue event -- which is
// what makes that step's (without communicator) or the step before's (with) output valid for the caller, in stream order.
// Round 3, later (opt-in, SP_DIST_TWO_LANES=1): the main kernels themselves go to two lanes of the engine's own (even / odd steps),
// each behind an event recorded on A just before (A's history = the caller's producer of x + the waits for the epilogues that free
// the scratch set): consecutive main kernels are then independent in the eyes of the hardware, and step k + 1's workgroups take
// the CUs as step k's retire.  Measured +0.8 % per step -- and NOT the default: two main kernels that share the chip have no
// duration of their own any more (rocprofv3 --kernel-trace reads 0.99 ms per launch instead of 0.54), which is what bench.py's
// roofline figure and its rocprof cross-check are built on.  x must stay valid until the step is reported.
struct EngineSlot {
    bool busy = false;
    std::vector<float> win;
    int64_t frames_total = 0;
    int sided = 0;
    double scale = 0;
    double *out = nullptr;
};
struct Engine {
    hipStream_t epi = nullptr;
    hipStream_t lane[2] = {nullptr, nullptr};          // SP_DIST_TWO_LANES=1: main kernels of even / odd steps (default: the launch stream)
    hipEvent_t ev_main[2] = {nullptr, nullptr}, ev_epi[2] = {nullptr, nullptr}, ev_in[2] = {nullptr, nullptr};
    Scratch work[2], onepass[2], trend[2], ticket[2], st[2];
    EngineSlot slot[2];
    int64_t nsub = 0;            // submits since the last flush
    bool with_comm = false;      // mode of the steps in flight
} geng;

int engine_init() {
    if (geng.epi) return 0;
    HIPCHK(hipStreamCreateWithFlags(&geng.epi, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        HIPCHK(hipStreamCreateWithFlags(&geng.lane[i], hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&geng.ev_main[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&geng.ev_epi[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&geng.ev_in[i], hipEventDisableTiming));
    }
    return 0;
}
void engine_release() {
    if (!geng.epi) return;
    (void)hipStreamSynchronize(geng.epi);
    for (int i = 0; i < 2; ++i) {
        if (geng.lane[i]) {
            (void)hipStreamSynchronize(geng.lane[i]);
            (void)hipStreamDestroy(geng.lane[i]);
            geng.lane[i] = nullptr;
        }
        (void)hipEventDestroy(geng.ev_main[i]);
        (void)hipEventDestroy(geng.ev_epi[i]);
        if (geng.ev_in[i]) (void)hipEventDestroy(geng.ev_in[i]);
        geng.ev_main[i] = geng.ev_epi[i] = geng.ev_in[i] = nullptr;
        geng.work[i].release();
        geng.onepass[i].release();
        geng.trend[i].release();
        geng.ticket[i].release();
        geng.st[i].release();
        geng.slot[i].busy = false;
    }
    (void)hipStreamDestroy(geng.epi);
    geng.epi = nullptr;
    geng.nsub = 0;
}

// apply the all-reduced state of slot s -> its output, on the engine's stream (B-ordered behind the collective)
int engine_apply_locked(int s) {
    EngineSlot &sl = geng.slot[s];
    const int nfft = (int)sl.win.size();
    Xf xf;
    if (get_xf(nfft, &xf)) return -1;
    void *Wf_d;
    if (get_window_spectrum(sl.win.data(), nfft, xf, &Wf_d)) return -1;
    LAUNCHCHK(launch_op_apply(LaunchCtx{geng.epi, g.ncu}, (const double *)geng.st[s].p, (const cf *)Wf_d, nfft, sl.sided,
                              sl.scale / (double)sl.frames_total, sl.out));
    return 0;
}

// this shard's additive state (sp_welch_export) into st_d (device, 5 nfft + 8 doubles)
int welch_export_locked(const void *xd, bool cplx, int64_t nsig, const float *win, int nfft, int hop, int64_t nframes,
                        int64_t nmean, double *st_d, const OpPrev *prev = nullptr, hipEvent_t prev_wait = nullptr,
                        bool *prev_done = nullptr, const SplitLaunch *sl = nullptr) {
    FusedOut fo{true, SP_SIDED_RAW, 1.0, st_d, false};
    if (prev) {
        fo.prev = *prev;
        fo.prev_wait = prev_wait;
    }
    if (welch_accum_locked(xd, cplx, nsig, win, nfft, hop, nframes, nmean, false, &fo, sl)) return -1;
    g_pend.valid = false;
    if (prev_done) *prev_done = fo.prev_done;
    if (fo.done) return 0;
    LAUNCHCHK(launch_op_finish(sl ? LaunchCtx{sl->epi, g.ncu} : lc(), g_pend.xd, g_pend.cplx, g_pend.trend_f, g_pend.win_d, g_pend.st, nullptr, g_pend.nmean,
                               g_pend.xf, g_pend.hop, g_pend.nframes, g_pend.cw, g_pend.Wf, SP_SIDED_RAW, 1.0, st_d, true));
    return 0;
}

}   // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

int sp_version(void) { return SP_VERSION; }
int sp_max_wg_fft(void) { return SP_MAX_WG_FFT; }
const char *sp_last_error(void) { return g_err.c_str(); }

int sp_init(int device_id) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (g.ready && (device_id < 0 || device_id == g.device)) return 0;
    if (g.ready)
        return fail("libspectral is bound to device %d for the life of the process (asked for %d)", g.device, device_id);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail("no HIP device visible (libspectral has no CPU fallback)");
    if (device_id < 0) {
        if (hipGetDevice(&device_id) != hipSuccess) device_id = 0;
    }
    if (device_id >= count) return fail("device %d out of range (%d visible)", device_id, count);
    HIPCHK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail("device %d is %s; libspectral is built for gfx950 only", device_id, prop.gcnArchName);
    g.device = device_id;
    g.ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    g.ready = true;
    return 0;
}

void sp_shutdown(void) {
    (void)sp_comm_destroy();
    std::lock_guard<std::mutex> lk(g.mu);
    if (!g.ready) return;
    engine_release();
    (void)hipDeviceSynchronize();
    for (auto &kv : g.twiddles) (void)hipFree(kv.second);
    g.twiddles.clear();
    for (auto &kv : g.blue) {
        (void)hipFree(kv.second.chirp);
        (void)hipFree(kv.second.bf);
    }
    g.blue.clear();
    g.in0.release();
    g.in1.release();
    g.out0.release();
    g.work.release();
    g.small.release();
    g.trends.release();
    g.onepass.release();
    g.ticket.release();
    g.epi.release();
    g.pend_trend.release();
    g.cmS.release();
    g.cmT.release();
    g.cmG.release();
    g.cmH.release();
    g.cmO.release();
    g.bigA.release();
    g.bigB.release();
    g.bigT.release();
    g.blueA.release();
    g.blueB.release();
    g.longrec.release();
    for (auto &kv : g.bigtw) {
        (void)hipFree((void *)kv.second.hi);
        (void)hipFree((void *)kv.second.lo);
    }
    g.bigtw.clear();
    for (auto &kv : g.blue_big) {
        (void)hipFree(kv.second.chirp);
        (void)hipFree(kv.second.bf);
    }
    g.blue_big.clear();
    g_pend.valid = false;
    tables_release();
    g.ready = false;
}

int sp_set_stream(void *hip_stream) {
    if (ensure_init()) return -1;
    ApiLock lk;
    hipStream_t ns = (hipStream_t)hip_stream;
    if (ns != g.stream) {
        // the library's scratch buffers are shared by all calls: work already queued on the old stream must finish
        // before work on the new one may reuse them.  (The old stream may have been destroyed by its owner: errors of
        // the record are dropped, there is nothing left to order against then.)
        if (!g.ev_sw && hipEventCreateWithFlags(&g.ev_sw, hipEventDisableTiming) != hipSuccess) g.ev_sw = nullptr;
        if (g.ev_sw) {
            if (hipEventRecord(g.ev_sw, g.stream) == hipSuccess) (void)hipStreamWaitEvent(ns, g.ev_sw, 0);
            (void)hipGetLastError();
        }
        g.stream = ns;
    }
    return 0;
}

int sp_synchronize(void) {
    if (ensure_init()) return -1;
    HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

int sp_profile_enable(int on) {
    if (ensure_init()) return -1;
    if (on && !g.ev0) {
        HIPCHK(hipEventCreate(&g.ev0));
        HIPCHK(hipEventCreate(&g.ev1));
    }
    g.profile = on != 0;
    g.prof_valid = false;
    return 0;
}

int sp_profile_last_ms(double *ms) {
    if (ensure_init()) return -1;
    if (!g.profile || !g.prof_valid) return fail("sp_profile_last_ms: no profiled launch recorded");
    HIPCHK(hipEventSynchronize(g.ev1));
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, g.ev0, g.ev1));
    *ms = (double)t;
    return 0;
}

const char *sp_profile_last_kernel(void) { return g.last_kernel; }

int sp_device_info(int64_t out[4]) {
    if (ensure_init()) return -1;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, g.device));
    out[0] = prop.multiProcessorCount;
    out[1] = (int64_t)prop.maxSharedMemoryPerMultiProcessor;
    out[2] = prop.clockRate;
    out[3] = prop.warpSize;
    return 0;
}

int sp_mean(const void *x, int x_dtype, int64_t n, double out[2], int mem) {
    if (ensure_init()) return -1;
    if (n <= 0) return fail("sp_mean: n must be positive");
    ApiLock lk;
    const size_t esz = x_dtype == SP_DTYPE_C64 ? 8 : 4;
    const void *xd = x;
    if (!mem) {
        if (g.in0.ensure(esz * (size_t)n)) return -1;
        HIPCHK(hipMemcpyAsync(g.in0.p, x, esz * (size_t)n, hipMemcpyHostToDevice, g.stream));
        xd = g.in0.p;
    }
    TrendBuf tb;
    if (get_trendbuf(1, &tb)) return -1;
    if (set_trend(tb, 0, xd, x_dtype == SP_DTYPE_C64, n, 1, 0, 0)) return -1;
    double h[2];
    HIPCHK(hipMemcpyAsync(h, tb.d, sizeof h, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    out[0] = h[0];
    out[1] = h[1];
    return 0;
}

int sp_fft_c2c(const void *in, void *out, int64_t n, int64_t batch, int direction, int mem) {
    if (ensure_init()) return -1;
    if (n < 1 || batch < 0) return fail("sp_fft_c2c: bad n/batch");
    if (direction != -1 && direction != 1) return fail("sp_fft_c2c: direction must be -1 or +1");
    if (batch == 0) return 0;
    ApiLock lk;
    const size_t bytes = sizeof(cf) * (size_t)n * (size_t)batch;
    const cf *din = (const cf *)in;
    cf *dout = (cf *)out;
    if (!mem) {
        if (g.in0.ensure(bytes)) return -1;
        HIPCHK(hipMemcpyAsync(g.in0.p, in, bytes, hipMemcpyHostToDevice, g.stream));
        din = (const cf *)g.in0.p;
        dout = (cf *)g.in0.p;
    }
    if (dev_fft_any(din, dout, n, batch, direction > 0)) return -1;
    if (!mem) {
        HIPCHK(hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

int sp_welch_psd(const void *x, int x_dtype, int64_t nsig, const float *win, int nfft, int hop, int64_t nframes,
                 int detrend, double mean_re, double mean_im, int sided, double scale, double *pxx_out, int mem) {
    if (ensure_init()) return -1;
    if (check_frames("sp_welch_psd", nsig, nfft, hop, nframes)) return -1;
    if (sided < 1 || sided > 3) return fail("sp_welch_psd: bad sided");
    if (detrend < 0 || detrend > 4) return fail("sp_welch_psd: detrend must be 0..4");
    const int segmean = detrend == SP_DETREND_SEGMEAN ? 1 : (detrend == SP_DETREND_SEGLINEAR ? 2 : 0);   // per-segment: generic kernel
    if (segmean) {
        detrend = SP_DETREND_CONST;
        mean_re = mean_im = 0.0;
    }
    ApiLock lk;
    const bool lng = !wg_capable(nfft);          // segment longer than one workgroup transform: multi-kernel path
    Xf xf;
    if (!lng && get_xf(nfft, &xf)) return -1;
    const bool cplx = x_dtype == SP_DTYPE_C64;
    const size_t esz = cplx ? 8 : 4;
    const void *xd = x;
    if (!mem) {
        if (g.in0.ensure(esz * (size_t)nsig)) return -1;
        HIPCHK(hipMemcpyAsync(g.in0.p, x, esz * (size_t)nsig, hipMemcpyHostToDevice, g.stream));
        xd = g.in0.p;
    }
    const int nb = nbins_host(nfft, sided);
    double *out_d = pxx_out;
    if (!mem) {
        if (g.out0.ensure(sizeof(double) * (size_t)nb)) return -1;
        out_d = (double *)g.out0.p;
    }
    const bool allow_carry = !env_flag("SP_WELCH_GENERIC");
    if (lng) {
        void *win_d;
        if (get_table(1, win, sizeof(float) * (size_t)nfft, &win_d, nullptr)) return -1;
        TrendBuf tb;
        if (get_trendbuf(1, &tb)) return -1;
        if (set_trend(tb, 0, xd, cplx, nsig, detrend, mean_re, mean_im)) return -1;
        const int64_t mc = long_chunk_frames(nfft, nframes);
        if (g.bigA.ensure(sizeof(cf) * (size_t)mc * (size_t)nfft) || g.work.ensure(sizeof(double) * (size_t)nfft)) return -1;
        cf *S = (cf *)g.bigA.p;
        double *acc = (double *)g.work.p;
        HIPCHK(hipMemsetAsync(acc, 0, sizeof(double) * (size_t)nfft, g.stream));
        for (int64_t f0 = 0; f0 < nframes; f0 += mc) {
            const int64_t m = nframes - f0 < mc ? nframes - f0 : mc;
            if (long_spectra(xd, cplx, (const float *)win_d, nfft, hop, f0, m, tb.f, detrend == 2, segmean, S, nullptr)) return -1;
            LAUNCHCHK(launch_long_acc_psd(lc(), S, m, nfft, acc));
        }
        LAUNCHCHK(launch_long_finish(lc(), acc, nfft, sided, scale / (double)nframes, false, out_d));
        g.last_kernel = "k_long_acc_psd";
    } else if (detrend == SP_DETREND_MEAN && allow_carry && !segmean && !env_flag("SP_WELCH_TWOPASS") &&
               welch_carry_eligible(xf, hop, false) &&
               (cplx || (2 * hop == nfft && nframes >= 2 && !env_flag("SP_NO_REALPAIR") &&
                         welch_pipe_wanted(xf, hop, (nframes + 1) / 2)))) {
        // global-mean detrend in ONE pass over the signal (estimate + exact correction in the epilogue)
        FusedOut fo{false, sided, scale / (double)nframes, out_d, false};
        if (welch_accum_locked(xd, cplx, nsig, win, nfft, hop, nframes, nsig, false, &fo)) return -1;
        if (!fo.done && welch_finish_locked(nullptr, nframes, sided, scale, out_d)) return -1;
    } else {
        void *win_d;
        if (get_table(1, win, sizeof(float) * (size_t)nfft, &win_d, nullptr)) return -1;
        TrendBuf tb;
        if (get_trendbuf(1, &tb)) return -1;
        if (set_trend(tb, 0, xd, cplx, nsig, detrend, mean_re, mean_im)) return -1;
        const bool pair = !cplx && nframes >= 2 && !segmean && !env_flag("SP_NO_REALPAIR");
        const bool pipe = !pair && !segmean && detrend != 2 && allow_carry && welch_pipe_wanted(xf, hop, nframes);
        const bool pipe_rp = pair && detrend != 2 && allow_carry && 2 * hop == nfft && welch_pipe_wanted(xf, hop, (nframes + 1) / 2);
        const RunPart rp = pipe_rp ? run_partition(xf.L, (nframes + 1) / 2, g.ncu, welch_pipe_gpc())
                                   : (pipe ? run_partition(xf.L, nframes, g.ncu, welch_pipe_gpc())
                                           : (pair ? run_partition(xf.L, (nframes + 1) / 2, g.ncu,
                                                                   getenv("SP_GROUPS_PER_CU") ? 0 : welch_rp_groups_per_cu(xf, detrend == 2))
                                                   : run_partition(xf.L, nframes, g.ncu)));
        if (g.work.ensure(sizeof(float) * (size_t)rp.groups * xf.L)) return -1;
        float *partial = (float *)g.work.p;
        {
            ProfScope ps;
            if (pipe_rp) {
                LAUNCHCHK(launch_welch_pipe(lc(), xd, false, (const float *)win_d, hop, nframes, tb.f, xf, partial, rp, nullptr, 3));
                g.last_kernel = "k_welch_pipe(realpair)";
            } else if (pair) {
                // real input: two frames per complex transform, |Z|^2 accumulated, symmetrised by the finish kernel
                LAUNCHCHK(launch_welch_rp(lc(), (const float *)xd, (const float *)win_d, hop, nframes, tb.f, detrend == 2, xf,
                                          partial, rp));
                g.last_kernel = "k_welch_rp";
            } else if (pipe) {
                LAUNCHCHK(launch_welch_pipe(lc(), xd, cplx, (const float *)win_d, hop, nframes, tb.f, xf, partial, rp, nullptr));
                g.last_kernel = "k_welch_pipe";
            } else {
                LAUNCHCHK(launch_welch(lc(), xd, cplx, (const float *)win_d, hop, nframes, tb.f, detrend == 2, xf, partial,
                                       rp, allow_carry, nullptr, &g.last_kernel, segmean));
            }
        }
        LAUNCHCHK(launch_welch_finish(lc(), partial, rp.groups, xf, sided, scale / (double)nframes, out_d, pair ? 1 : 0));
    }
    if (!mem) {
        HIPCHK(hipMemcpyAsync(pxx_out, out_d, sizeof(double) * (size_t)nb, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

int sp_welch_accum(const void *x, int x_dtype, int64_t nsig, const float *win, int nfft, int hop, int64_t nframes,
                   int64_t nmean, double *sum_out, int mem) {
    if (ensure_init()) return -1;
    if (check_frames("sp_welch_accum", nsig, nfft, hop, nframes)) return -1;
    ApiLock lk;
    const bool cplx = x_dtype == SP_DTYPE_C64;
    const size_t esz = cplx ? 8 : 4;
    const void *xd = x;
    if (!mem) {
        if (g.in0.ensure(esz * (size_t)nsig)) return -1;
        HIPCHK(hipMemcpyAsync(g.in0.p, x, esz * (size_t)nsig, hipMemcpyHostToDevice, g.stream));
        xd = g.in0.p;
    }
    if (welch_accum_locked(xd, cplx, nsig, win, nfft, hop, nframes, nmean, true)) return -1;
    if (sum_out) {
        if (mem) HIPCHK(hipMemcpyAsync(sum_out, g_pend.sum_d, sizeof(double) * 2, hipMemcpyDeviceToDevice, g.stream));
        else {
            HIPCHK(hipMemcpyAsync(sum_out, g_pend.sum_d, sizeof(double) * 2, hipMemcpyDeviceToHost, g.stream));
            HIPCHK(hipStreamSynchronize(g.stream));
        }
    }
    return 0;
}

int sp_welch_finish(const double *mean, int64_t frames_total, int sided, double scale, double *pxx_out, int mem) {
    if (ensure_init()) return -1;
    if (sided < 1 || sided > 3) return fail("sp_welch_finish: bad sided");
    if (frames_total < 1) return fail("sp_welch_finish: frames_total must be positive");
    ApiLock lk;
    if (!g_pend.valid) return fail("sp_welch_finish: no pending sp_welch_accum");
    const int nb = nbins_host(g_pend.nfft, sided);
    double *out_d = pxx_out;
    const double *mean_d = mean;
    if (!mem) {
        if (g.out0.ensure(sizeof(double) * (size_t)nb + 64)) return -1;
        out_d = (double *)g.out0.p;
        if (mean) {
            double *md = out_d + nb + 2;
            HIPCHK(hipMemcpyAsync(md, mean, sizeof(double) * 2, hipMemcpyHostToDevice, g.stream));
            mean_d = md;
        }
    }
    if (welch_finish_locked(mean_d, frames_total, sided, scale, out_d)) return -1;
    if (!mem) {
        HIPCHK(hipMemcpyAsync(pxx_out, out_d, sizeof(double) * (size_t)nb, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

int sp_welch_export(const void *x, int x_dtype, int64_t nsig, const float *win, int nfft, int hop, int64_t nframes,
                    int64_t nmean, double *state, int mem) {
    if (ensure_init()) return -1;
    if (check_frames("sp_welch_export", nsig, nfft, hop, nframes)) return -1;
    ApiLock lk;
    const bool cplx = x_dtype == SP_DTYPE_C64;
    const size_t esz = cplx ? 8 : 4;
    const void *xd = x;
    if (!mem) {
        if (g.in0.ensure(esz * (size_t)nsig)) return -1;
        HIPCHK(hipMemcpyAsync(g.in0.p, x, esz * (size_t)nsig, hipMemcpyHostToDevice, g.stream));
        xd = g.in0.p;
    }
    const size_t nst = 5 * (size_t)nfft + 8;
    double *st_d = state;
    if (!mem) {
        if (g.out0.ensure(sizeof(double) * nst)) return -1;
        st_d = (double *)g.out0.p;
    }
    if (welch_export_locked(xd, cplx, nsig, win, nfft, hop, nframes, nmean, st_d)) return -1;
    if (!mem) {
        HIPCHK(hipMemcpyAsync(state, st_d, sizeof(double) * nst, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

int sp_welch_apply(const double *state, const float *win, int nfft, int64_t frames_total, int sided, double scale,
                   double *pxx_out, int mem) {
    if (ensure_init()) return -1;
    if (sided < 1 || sided > 3) return fail("sp_welch_apply: bad sided");
    if (frames_total < 1) return fail("sp_welch_apply: frames_total must be positive");
    ApiLock lk;
    if (!wg_capable(nfft))
        return fail("sp_welch_apply: segments longer than one workgroup transform are not sharded (nfft = %d, at most %d)", nfft,
                    SP_MAX_WG_FFT);
    Xf xf;
    if (get_xf(nfft, &xf)) return -1;
    if (xf.blue) return fail("sp_welch_apply: power-of-two nfft only");
    void *Wf_d;
    if (get_window_spectrum(win, nfft, xf, &Wf_d)) return -1;
    const int nb = nbins_host(nfft, sided);
    const size_t nst = 5 * (size_t)nfft + 8;
    const double *st_d = state;
    double *out_d = pxx_out;
    if (!mem) {
        if (g.out0.ensure(sizeof(double) * (nst + (size_t)nb) + 64)) return -1;
        double *sd = (double *)g.out0.p;
        HIPCHK(hipMemcpyAsync(sd, state, sizeof(double) * nst, hipMemcpyHostToDevice, g.stream));
        st_d = sd;
        out_d = sd + nst;
    }
    LAUNCHCHK(launch_op_apply(lc(), st_d, (const cf *)Wf_d, nfft, sided, scale / (double)frames_total, out_d));
    if (!mem) {
        HIPCHK(hipMemcpyAsync(pxx_out, out_d, sizeof(double) * (size_t)nb, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

/* ---- multi-GPU inside the library: one process per GPU, RCCL over xGMI --------------------------------------------------- */
int sp_comm_unique_id(void *id_out) {
    if (!id_out) return fail("sp_comm_unique_id: null pointer");
    std::lock_guard<std::mutex> lk(g.mu);
    if (rccl_load()) return -1;
    static_assert(sizeof(ncclUniqueId) == SP_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    NCCLCHK(rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return 0;
}

int sp_comm_init(const void *id_in, int world, int rank) {
    if (ensure_init()) return -1;
    if (!id_in || world < 1 || rank < 0 || rank >= world) return fail("sp_comm_init: bad id/world/rank");
    ApiLock lk;
    if (gcomm.comm) return fail("sp_comm_init: a communicator exists already (sp_comm_destroy first)");
    if (geng.nsub) return fail("sp_comm_init: streaming steps are in flight (sp_welch_dist_flush first)");
    if (rccl_load()) return -1;
    HIPCHK(hipSetDevice(g.device));
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof id);
    // at most `ctas` workgroups for this communicator's kernels (ncclConfig_t in its first public layout, NCCL 2.17: newer
    // libraries accept the shorter structure by its size / version fields); without the entry point: RCCL's own choice
    struct ConfigV217 {
        size_t size;
        unsigned magic, version;
        int blocking, cgaClusterSize, minCTAs, maxCTAs;
        const char *netName;
    };
    const int ctas = dist_rccl_ctas();
    gcomm.ctas = 0;
    if (rccl.CommInitRankConfig && ctas > 0) {
        ConfigV217 cfg{sizeof(ConfigV217), 0xcafebeefu, 21700u, INT_MIN, INT_MIN, 1, ctas, nullptr};
        NCCLCHK(rccl.CommInitRankConfig(&gcomm.comm, world, id, rank, &cfg));
        gcomm.ctas = ctas;
    } else {
        NCCLCHK(rccl.CommInitRank(&gcomm.comm, world, id, rank));
    }
    gcomm.world = world;
    gcomm.rank = rank;
    return 0;
}

int sp_comm_info(int out[2]) {
    out[0] = gcomm.comm ? gcomm.world : 0;
    out[1] = gcomm.comm ? gcomm.rank : -1;
    return 0;
}

int sp_comm_destroy(void) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (!gcomm.comm) return 0;
    if (geng.epi) (void)hipStreamSynchronize(geng.epi);
    (void)hipStreamSynchronize(g.stream);
    (void)rccl.CommDestroy(gcomm.comm);
    gcomm.comm = nullptr;
    gcomm.world = 0;
    gcomm.rank = -1;
    geng.nsub = 0;
    geng.slot[0].busy = geng.slot[1].busy = false;
    return 0;
}

int sp_welch_dist_submit(const void *x, int x_dtype, int64_t nsig, const float *win, int nfft, int hop, int64_t nframes,
                         int64_t nmean, int64_t frames_total, int sided, double scale, double *pxx_out, int *ndone) {
    if (ensure_init()) return -1;
    if (check_frames("sp_welch_dist_submit", nsig, nfft, hop, nframes)) return -1;
    if (sided < 1 || sided > 3) return fail("sp_welch_dist_submit: bad sided");
    if (frames_total < nframes) return fail("sp_welch_dist_submit: frames_total is the frame count of the WHOLE stream");
    if (!x || !pxx_out) return fail("sp_welch_dist_submit: null pointer");
    if (ndone) *ndone = 0;
    ApiLock lk;
    if (engine_init()) return -1;
    const bool comm = gcomm.comm != nullptr;
    if (geng.nsub && comm != geng.with_comm) return fail("sp_welch_dist_submit: communicator changed with steps in flight");
    geng.with_comm = comm;
    const bool cplx = x_dtype == SP_DTYPE_C64;
    const int64_t k = geng.nsub;
    const int s = (int)(k & 1), o = 1 - s;
    // slot s held step k - 2: its epilogue event was waited for on the launch stream during submit k - 1 (below), so the main
    // kernel of this step may overwrite its scratch set
    EngineSlot &cur = geng.slot[s], &prv = geng.slot[o];
    SplitLaunch sl{&geng.work[s], &geng.onepass[s], &geng.trend[s], &geng.ticket[s], geng.epi, geng.ev_main[s]};
    sl.reserve_cus = comm ? dist_reserved_cus() : 0;
    if (env_flag("SP_DIST_TWO_LANES")) {          // opt-in (see the engine's comment): overlapping main kernels have no duration of their own
        sl.main = geng.lane[s];
        sl.ev_in = geng.ev_in[s];
    }
    const LaunchCtx ec{geng.epi, g.ncu};
    if (comm) {
        const size_t nst = 5 * (size_t)nfft + 8;
        if (geng.st[s].ensure(sizeof(double) * nst)) return -1;
        double *st_d = (double *)geng.st[s].p;
        // the launch that finishes THIS step's state also applies the previous step's all-reduced state (same transform length;
        // B-ordered behind that step's collective) -- otherwise a k_op_apply launch of its own, on B as well
        OpPrev prev{nullptr, nullptr, nullptr, 0, 0.0};
        if (prv.busy && (int)prv.win.size() == nfft && !env_flag("SP_DIST_SEPARATE_APPLY")) {
            Xf xfo;
            if (get_xf(nfft, &xfo)) return -1;
            void *Wf_o;
            if (get_window_spectrum(prv.win.data(), nfft, xfo, &Wf_o)) return -1;
            prev = OpPrev{(const double *)geng.st[o].p, (const cf *)Wf_o, prv.out, prv.sided, prv.scale / (double)prv.frames_total};
        }
        bool prev_done = false;
        if (welch_export_locked(x, cplx, nsig, win, nfft, hop, nframes, nmean, st_d, prev.st ? &prev : nullptr, nullptr, &prev_done, &sl))
            return -1;
        if (prv.busy && !prev_done && engine_apply_locked(o)) return -1;
        NCCLCHK(rccl.AllReduce(st_d, st_d, nst, ncclDouble, ncclSum, gcomm.comm, geng.epi));
    } else {
        FusedOut fo{false, sided, scale / (double)frames_total, pxx_out, false};
        if (welch_accum_locked(x, cplx, nsig, win, nfft, hop, nframes, nmean, false, &fo, &sl)) return -1;
        if (!fo.done && welch_finish_locked(nullptr, frames_total, sided, scale, pxx_out, &ec)) return -1;
    }
    HIPCHK(hipEventRecord(geng.ev_epi[s], geng.epi));
    cur.busy = true;
    cur.win.assign(win, win + nfft);
    cur.frames_total = frames_total;
    cur.sided = sided;
    cur.scale = scale;
    cur.out = pxx_out;
    geng.nsub = k + 1;
    // the previous step's epilogue has had this step's main kernel to run beside: wait for it on the launch stream NOW (behind
    // the main kernel just enqueued).  That makes valid, in stream order: without communicator the output of step k - 1; with
    // one the output of step k - 2 (applied by step k - 1's epilogue launch).
    if (prv.busy) {
        // (this wait is a packet of its own between two main kernels: ~4 us per step, measured by leaving it out -- 0.0826 ->
        //  0.0784 ms at 2^25 samples.  It is what orders the scratch reuse and validates outputs in stream order; waiting only every
        //  few steps over more scratch sets would save most of it at the price of results arriving in batches: not done)
        HIPCHK(hipStreamWaitEvent(g.stream, geng.ev_epi[o], 0));
        if (!comm) {
            prv.busy = false;
            if (ndone) *ndone = 1;
        } else if (k >= 2 && ndone) {
            *ndone = 1;
        }
    }
    return 0;
}

int sp_welch_dist_flush(int *ndone) {
    if (ensure_init()) return -1;
    if (ndone) *ndone = 0;
    ApiLock lk;
    const int64_t k = geng.nsub;
    if (k == 0 || !geng.epi) return 0;
    const int s = (int)((k - 1) & 1);                 // the slot submitted last
    int done = 1;
    if (geng.with_comm) {
        // the last step's state still waits for its apply (B-ordered behind its collective); the step before it was applied by the
        // last step's epilogue launch and becomes valid with the same wait
        if (engine_apply_locked(s)) return -1;
        HIPCHK(hipEventRecord(geng.ev_epi[s], geng.epi));
        if (k >= 2) done = 2;
    }
    HIPCHK(hipStreamWaitEvent(g.stream, geng.ev_epi[s], 0));
    geng.slot[0].busy = geng.slot[1].busy = false;
    geng.nsub = 0;
    if (ndone) *ndone = done;
    return 0;
}

int sp_welch_psd_dist(const void *x, int x_dtype, int64_t nsig, const float *win, int nfft, int hop, int64_t nframes,
                      int64_t nmean, int64_t frames_total, int sided, double scale, double *pxx_out) {
    {
        std::lock_guard<std::mutex> lk(g.mu);
        if (geng.nsub) return fail("sp_welch_psd_dist: streaming steps are in flight (sp_welch_dist_flush first)");
    }
    int n = 0;
    if (sp_welch_dist_submit(x, x_dtype, nsig, win, nfft, hop, nframes, nmean, frames_total, sided, scale, pxx_out, &n)) return -1;
    return sp_welch_dist_flush(&n);
}

int sp_welch_csd(const void *x, const void *y, int dtype, int64_t nsig, int nch, int64_t y_ld, const float *win,
                 int nfft, int hop, int64_t nframes, int detrend, const double *mean_x, const double *mean_y,
                 int sided, double scale, double *pxx, double *pyy, double *pxy, int mem) {
    if (ensure_init()) return -1;
    if (check_frames("sp_welch_csd", nsig, nfft, hop, nframes)) return -1;
    if (nch < 1 || y_ld < nsig) return fail("sp_welch_csd: bad nch / y_ld");
    if (sided < 1 || sided > 3) return fail("sp_welch_csd: bad sided");
    if (detrend < 0 || detrend > 4) return fail("sp_welch_csd: detrend must be 0..4");
    const int segmean = detrend == SP_DETREND_SEGMEAN ? 1 : (detrend == SP_DETREND_SEGLINEAR ? 2 : 0);
    if (segmean) {
        detrend = SP_DETREND_CONST;
        mean_x = mean_y = nullptr;
    }
    ApiLock lk;
    const bool lng = !wg_capable(nfft);
    Xf xf;
    if (!lng && get_xf(nfft, &xf)) return -1;
    const bool cplx = dtype == SP_DTYPE_C64;
    const size_t esz = cplx ? 8 : 4;
    const void *xd = x, *yd = y;
    if (!mem) {
        if (g.in0.ensure(esz * (size_t)nsig)) return -1;
        if (g.in1.ensure(esz * (size_t)y_ld * (size_t)nch)) return -1;
        HIPCHK(hipMemcpyAsync(g.in0.p, x, esz * (size_t)nsig, hipMemcpyHostToDevice, g.stream));
        HIPCHK(hipMemcpyAsync(g.in1.p, y, esz * (size_t)y_ld * (size_t)nch, hipMemcpyHostToDevice, g.stream));
        xd = g.in0.p;
        yd = g.in1.p;
    }
    void *win_d;
    if (get_table(1, win, sizeof(float) * (size_t)nfft, &win_d, nullptr)) return -1;
    TrendBuf tb;
    if (get_trendbuf(nch + 1, &tb)) return -1;
    if (set_trend(tb, 0, xd, cplx, nsig, detrend, mean_x ? mean_x[0] : 0, mean_x ? mean_x[1] : 0)) return -1;
    // real x against many real channels: the pair path below; with mean detrend at nfft 4096 / 50 % overlap the channels' means
    // are not taken by a pass of their own (4.2 GB at 63 channels x 2^24: 0.7 of 2.9 ms) but in one pass with the spectra
    // (k_welch_csd_pair<OP>: estimate + block sums, exact correction in k_csd_pair_finish); SP_CSD_TWOPASS=1: the separate pass
    const bool pair_path = !lng && !cplx && !segmean && csd_rp_eligible(xf) && nch >= 2 && nframes >= 2 && !env_flag("SP_NO_REALPAIR") &&
                           !env_flag("SP_CSD_XIY");
    // (from 8 channels on: below, the epilogue's fixed ~0.1 ms costs more than the channels' pass, 0.39 against 0.35 ms at 2 channels
    //  x 2^24, 0.55 / 0.56 at 8, 0.76 / 0.83 at 16; SP_CSD_ONEPASS=1 forces the form)
    const bool pair_op = pair_path && detrend == SP_DETREND_MEAN && xf.L == 4096 && 2 * hop == xf.L && nch <= 512 &&
                         (nch >= 8 || env_flag("SP_CSD_ONEPASS")) && !env_flag("SP_CSD_TWOPASS");
    if (pair_op) {
        HIPCHK(hipMemsetAsync(tb.f + 4, 0, sizeof(float) * 4 * (size_t)nch, g.stream));     // (the kernel publishes its estimates here)
    } else if (detrend != 0 && nch <= 512) {
        // all channels' means / trend lines in one launch pair (64 channels one by one cost 3 ms of launches)
        double *scr = moments_scratch();
        if (!scr) return -1;
        LAUNCHCHK(launch_moments(lc(), yd, cplx, nsig, detrend, scr, tb.d + 8, tb.f + 4, nch, y_ld));
    } else {
        for (int c = 0; c < nch; ++c)
            if (set_trend(tb, c + 1, (const char *)yd + esz * (size_t)y_ld * (size_t)c, cplx, nsig, detrend,
                          mean_y ? mean_y[2 * c] : 0, mean_y ? mean_y[2 * c + 1] : 0))
                return -1;
    }
    const size_t nb = (size_t)nbins_host(nfft, sided);
    double *pxx_d = pxx, *pyy_d = pyy, *pxy_d = pxy;
    if (!mem) {
        if (g.out0.ensure(sizeof(double) * nb * (1 + 3 * (size_t)nch))) return -1;
        pxx_d = (double *)g.out0.p;
        pyy_d = pxx_d + nb;
        pxy_d = pyy_d + nb * nch;
    }
    if (lng) {
        // long segments: spectra of a chunk of frames of x, then of every channel against them (k_long.hip)
        const int64_t mc = long_chunk_frames(nfft, nframes);
        const size_t nf = (size_t)nfft;
        if (g.bigA.ensure(sizeof(cf) * (size_t)mc * nf) || g.bigB.ensure(sizeof(cf) * (size_t)mc * nf) ||
            g.work.ensure(sizeof(double) * nf * (1 + 3 * (size_t)nch)))
            return -1;
        cf *Sx = (cf *)g.bigA.p, *Sy = (cf *)g.bigB.p;
        double *axx = (double *)g.work.p, *ayy = axx + nf, *axy = ayy + nf * (size_t)nch;
        HIPCHK(hipMemsetAsync(axx, 0, sizeof(double) * nf * (1 + 3 * (size_t)nch), g.stream));
        for (int64_t f0 = 0; f0 < nframes; f0 += mc) {
            const int64_t m = nframes - f0 < mc ? nframes - f0 : mc;
            if (long_spectra(xd, cplx, (const float *)win_d, nfft, hop, f0, m, tb.f, detrend == 2, segmean, Sx, nullptr)) return -1;
            LAUNCHCHK(launch_long_acc_psd(lc(), Sx, m, nfft, axx));
            for (int c = 0; c < nch; ++c) {
                if (long_spectra((const char *)yd + esz * (size_t)y_ld * (size_t)c, cplx, (const float *)win_d, nfft, hop, f0, m,
                                 tb.f + 4 * (c + 1), detrend == 2, segmean, Sy, nullptr))
                    return -1;
                LAUNCHCHK(launch_long_acc_csd(lc(), Sx, Sy, m, nfft, ayy + nf * (size_t)c, axy + 2 * nf * (size_t)c));
            }
        }
        const double sc = scale / (double)nframes;
        LAUNCHCHK(launch_long_finish(lc(), axx, nfft, sided, sc, false, pxx_d));
        for (int c = 0; c < nch; ++c) {
            LAUNCHCHK(launch_long_finish(lc(), ayy + nf * (size_t)c, nfft, sided, sc, false, pyy_d + nb * (size_t)c));
            LAUNCHCHK(launch_long_finish(lc(), axy + 2 * nf * (size_t)c, nfft, sided, sc, true, pxy_d + 2 * nb * (size_t)c));
        }
        if (!mem) {
            HIPCHK(hipMemcpyAsync(pxx, pxx_d, sizeof(double) * nb, hipMemcpyDeviceToHost, g.stream));
            HIPCHK(hipMemcpyAsync(pyy, pyy_d, sizeof(double) * nb * nch, hipMemcpyDeviceToHost, g.stream));
            HIPCHK(hipMemcpyAsync(pxy, pxy_d, sizeof(double) * nb * nch * 2, hipMemcpyDeviceToHost, g.stream));
            HIPCHK(hipStreamSynchronize(g.stream));
        }
        return 0;
    }
    const RunPart rp = run_partition_2d(xf.L, nframes, g.ncu, nch);
    if (g.work.ensure(sizeof(float) * (size_t)rp.groups * xf.L * 4 * (size_t)nch)) return -1;
    float *partial = (float *)g.work.p;
    if (pair_path) {
        // real x against many real channels: the reference's packed pair spectra once, then one transform per
        // (channel, frame PAIR); Pxx from the real-pair PSD kernel.  (SP_CSD_XIY=1: the x + i y_c form below.)
        const int64_t npairs = (nframes + 1) / 2;
        const RunPart rpp = run_partition_2d(xf.L, npairs, g.ncu, nch, (int64_t)g.ncu * csd_pair_resident(xf, detrend == 2, pair_op));
        if (g.work.ensure(sizeof(float) * (size_t)rpp.groups * xf.L * 4 * (size_t)nch)) return -1;
        partial = (float *)g.work.p;
        if (g.cmS.ensure(sizeof(cf) * (size_t)npairs * (size_t)xf.L)) return -1;
        cf *Zx = (cf *)g.cmS.p;
        const RunPart rpx = run_partition(xf.L, npairs, g.ncu);
        LAUNCHCHK(launch_pairspec(lc(), (const float *)xd, (const float *)win_d, hop, nframes, tb.f, detrend == 2, xf, rpx, Zx));
        if (pair_op) {
            const size_t N5 = (size_t)(5 * nfft + 8);
            const size_t b_spy = sizeof(cf) * (size_t)nch * (size_t)rpp.groups * (size_t)hop;
            const size_t b_sp = b_spy + sizeof(cf) * SP_COLSUM_SLICES * (size_t)hop;          // + the reference's slice sums
            const size_t b_sl = sizeof(double) * 2 * (size_t)hop * (size_t)(nch + 1), b_st = sizeof(double) * N5 * (size_t)(nch + 1);
            if (g.cmO.ensure(b_sp + b_sl + b_st)) return -1;
            cf *spartial = (cf *)g.cmO.p;
            double *Sl = (double *)((char *)g.cmO.p + b_sp), *Slx = Sl + 2 * (size_t)hop * (size_t)nch;
            double *st = (double *)((char *)g.cmO.p + b_sp + b_sl), *stx = st + N5 * (size_t)nch;
            void *Wf_d;
            if (get_window_spectrum(win, nfft, xf, &Wf_d)) return -1;
            LAUNCHCHK(launch_csd_pair(lc(), (const float *)yd, nch, y_ld, (const float *)win_d, hop, nframes, tb.f + 4, false, xf, Zx,
                                      partial, rpp, spartial));
            LAUNCHCHK(launch_cm_blocksums(lc(), spartial, nch, (int)rpp.groups, hop, Sl));
            LAUNCHCHK(launch_op_finish_channels(lc(), (const float *)yd, y_ld, nch, tb.f + 4, (const float *)win_d, Sl, (const cf *)Wf_d,
                                                hop, nframes, nsig, xf, st));
            // the reference (detrended by its exact mean: its d is rounding only, but B_x is needed for the channels' terms)
            cf *spx = (cf *)((char *)g.cmO.p + b_spy);
            LAUNCHCHK(launch_colsum_real(lc(), (const float *)xd, tb.f, hop, nframes, spx));
            LAUNCHCHK(launch_cm_blocksums(lc(), spx, 1, SP_COLSUM_SLICES, hop, Slx));
            LAUNCHCHK(launch_op_finish_channels(lc(), (const float *)xd, 0, 1, tb.f, (const float *)win_d, Slx, (const cf *)Wf_d, hop,
                                                nframes, nsig, xf, stx));
            LAUNCHCHK(launch_csd_pair_finish(lc(), partial, rpp.groups, xf, nch, sided, scale / (double)nframes, pyy_d, pxy_d, st, stx,
                                             (const cf *)Wf_d, tb.f, tb.f + 4, nsig, nframes));
        } else {
            LAUNCHCHK(launch_csd_pair(lc(), (const float *)yd, nch, y_ld, (const float *)win_d, hop, nframes, tb.f + 4, detrend == 2,
                                      xf, Zx, partial, rpp));
            LAUNCHCHK(launch_csd_pair_finish(lc(), partial, rpp.groups, xf, nch, sided, scale / (double)nframes, pyy_d, pxy_d));
        }
        // Pxx: real-pair Welch PSD of x (its own small partial buffer after the channels' one)
        if (g.bigT.ensure(sizeof(float) * (size_t)rpx.groups * xf.L)) return -1;
        float *px = (float *)g.bigT.p;
        LAUNCHCHK(launch_welch_rp(lc(), (const float *)xd, (const float *)win_d, hop, nframes, tb.f, detrend == 2, xf, px, rpx));
        LAUNCHCHK(launch_welch_finish(lc(), px, rpx.groups, xf, sided, scale / (double)nframes, pxx_d, 1));
    } else if (!cplx && !segmean && csd_rp_eligible(xf) && !env_flag("SP_NO_REALPAIR")) {
        // real x, y: x + i y_c in one transform per (frame, channel)
        LAUNCHCHK(launch_csd_rp(lc(), (const float *)xd, (const float *)yd, nch, y_ld, (const float *)win_d, hop, nframes,
                                tb.f, tb.f + 4, detrend == 2, xf, partial, rp));
        LAUNCHCHK(launch_csd_rp_finish(lc(), partial, rp.groups, xf, nch, sided, scale / (double)nframes, pxx_d, pyy_d,
                                       pxy_d));
    } else {
        LAUNCHCHK(launch_csd(lc(), xd, yd, cplx, nch, y_ld, (const float *)win_d, hop, nframes, tb.f, tb.f + 4,
                             detrend == 2, xf, partial, rp, segmean));
        LAUNCHCHK(launch_csd_finish(lc(), partial, rp.groups, xf, nch, sided, scale / (double)nframes, pxx_d, pyy_d,
                                    pxy_d));
    }
    if (!mem) {
        HIPCHK(hipMemcpyAsync(pxx, pxx_d, sizeof(double) * nb, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipMemcpyAsync(pyy, pyy_d, sizeof(double) * nb * nch, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipMemcpyAsync(pxy, pxy_d, sizeof(double) * nb * nch * 2, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

__global__ void k_set_trends(const double *__restrict__ means, float *__restrict__ trend, int nch) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < nch) {
        trend[4 * c + 0] = (float)means[c];
        trend[4 * c + 1] = 0.f;
        trend[4 * c + 2] = 0.f;
        trend[4 * c + 3] = 0.f;
    }
}

static int csd_matrix_impl(const char *who, const float *x, int nch, int64_t nsig, int64_t x_ld, const float *win, int nfft,
                           int hop, int64_t nframes, int detrend, const double *means_host, double scale, double *g_out,
                           int mem) {
    if (ensure_init()) return -1;
    if (check_frames(who, nsig, nfft, hop, nframes)) return -1;
    if (nch < 1 || x_ld < nsig) return fail("%s: bad nch / x_ld", who);
    if (detrend < 0 || detrend > 2) return fail("%s: detrend must be 0, 1 or 2", who);
    ApiLock lk;
    const bool lng = !wg_capable(nfft);          // segments longer than one workgroup transform: spectra by the long path
    Xf xf;
    xf.L = 0;
    xf.blue = true;
    if (!lng && get_xf(nfft, &xf)) return -1;
    const int nb = nfft / 2 + 1;
    const float *xd = x;
    if (!mem) {
        const size_t ib = sizeof(float) * (size_t)x_ld * (size_t)nch;
        if (g.in0.ensure(ib)) return -1;
        HIPCHK(hipMemcpyAsync(g.in0.p, x, ib, hipMemcpyHostToDevice, g.stream));
        xd = (const float *)g.in0.p;
    }
    void *win_d;
    if (get_table(1, win, sizeof(float) * (size_t)nfft, &win_d, nullptr)) return -1;
    TrendBuf tb;
    if (get_trendbuf(2 * nch, &tb)) return -1;           // second half: trends re-based to the current frame chunk
    const size_t gbytes = sizeof(double) * 2 * (size_t)nb * (size_t)nch * (size_t)nch;
    double *G = g_out;
    if (!mem) {
        if (g.cmG.ensure(gbytes)) return -1;
        G = (double *)g.cmG.p;
    }
    // (G is zeroed lazily, by the first chunk that adds into it directly: a call whose chunks all go through the packed spectra
    //  never needs it -- k_csdm_fold then stores, scales and mirrors in its one sweep)
    bool g_zeroed = false;
    // frames are processed in chunks so the two spectra buffers stay <= 8 GiB each (and float sums stay short)
    int64_t mc = ((int64_t)1 << 31) / ((int64_t)nch * nb);     // <= 16 GiB per spectra buffer
    if (mc > 16384) mc = 16384;
    if (mc < 32) mc = 32;
    mc &= ~(int64_t)31;
    {
        // equal chunks (a ragged last chunk of a few frames costs whole launches)
        const int64_t nchunks = (nframes + mc - 1) / mc;
        mc = (nframes + nchunks - 1) / nchunks;
        mc = (mc + 31) & ~(int64_t)31;
    }
    if (lng) {
        const int64_t lm = long_chunk_frames(nfft, nframes);
        if (mc > lm) mc = lm;
    }
    if (const char *e = getenv("SP_CSDM_CHUNK")) {               // experiment: frames per chunk (a multiple of 32)
        const int64_t v = atoll(e) & ~(int64_t)31;
        if (v >= 32 && v < mc) mc = v;
    }
    if (mc > nframes) mc = nframes;
    // contraction on the matrix cores: fused form reading the STFT output as it lies (default), the form with a
    // transposed copy (SP_CSDM_TRANSPOSED=1), or the VALU kernel (SP_CSDM_VALU=1); the last two are kept for A/B tests
    const bool use_mfma = !env_flag("SP_CSDM_VALU");
    const bool use_fused = use_mfma && nch <= 64 && !env_flag("SP_CSDM_TRANSPOSED");   // off-diagonal superblocks need 128 accumulators
    const int nchp = (nch + 63) / 64 * 64;                       // MFMA layout: channels padded to whole 64-superblocks,
    const int64_t mcp = (mc + 31) / 32 * 32;                     // frames to a multiple of 32 (zero filled)
    // fused path: every (channel, frame) row of the spectra starts on a 128-byte line (row pitch padded to a multiple of 16
    // bins).  With the natural pitch nb = nfft/2 + 1 (odd) almost every 16-bin tile row straddled two lines and the
    // contraction fetched 1.9x its algorithmic bytes (calibrated PMC: profiles/r02_fetch_size_calibration.txt)
    const bool rp_stft = !lng && !xf.blue && xf.L >= 32 && nch <= 65535;
    // bf16-split contraction (k_csdm_bf16.hip): needs the pair-interleaved spectra the real-pair STFT kernel can write
    const bool use_bf16 = use_fused && rp_stft && !env_flag("SP_CSDM_FP32");
    // two bf16 pieces per operand (16 bits, k_csdm_bf16<4>: 10 MFMAs per two frame pairs instead of 16) from 1024 frame pairs
    // on: there the float32 accumulation bounds the accuracy of both forms alike (1.6e-6 of the peak at 2049 frames against
    // the float64 oracle, tools/split2_ab.py) and the operand rounding (rms 2^-17/sqrt 3 per value, zero mean) averages to
    // <= 2e-7.  SP_CSDM_SPLIT3=1: three pieces always; SP_CSDM_SPLIT2=1: two pieces always
    const int split2 = env_flag("SP_CSDM_SPLIT2") ? 1 : (env_flag("SP_CSDM_SPLIT3") ? 0 : ((nframes + 1) / 2 >= 1024 ? 1 : 0));
    const int ld = use_bf16 ? (nb + 7) / 8 * 8
                            : ((use_fused && (rp_stft || lng) && !env_flag("SP_CSDM_NOPAD")) ? (nb + 15) / 16 * 16 : nb);
    const size_t sbytes = use_bf16 ? sizeof(cf) * 2 * (size_t)64 * (size_t)((mc + 1) / 2) * (size_t)ld       // 64 channel slots
                                   : sizeof(cf) * (size_t)nch * (size_t)mc * (size_t)ld;
    const size_t tbytes = use_fused ? sizeof(cf) * (size_t)nchp * (size_t)mcp * 16
                                    : (use_mfma ? sizeof(cf) * (size_t)nchp * (size_t)mcp * (size_t)nb : sbytes);
    if (g.cmS.ensure(sbytes) || g.cmT.ensure(tbytes)) return -1;
    cf *Xs = (cf *)g.cmS.p, *Xt = (cf *)g.cmT.p;
    // nfft 4096 at 50 % overlap: spectra by the pipeline of specialised waves (see below); m = frames of a chunk
    auto pipe_spec = [&](int64_t m) {
        static const int64_t minp = getenv("SP_CSDM_MINPAIRS") ? atoll(getenv("SP_CSDM_MINPAIRS")) : 32;    // pairs per run
        return use_bf16 && nfft == 4096 && 2 * hop == nfft && detrend != 2 && (m + 1) / 2 >= minp * (int64_t)((g.ncu + nch - 1) / nch) &&
               welch_pipe_wanted(xf, hop, (int64_t)1 << 40) && !env_flag("SP_CSDM_NOPIPESPEC");
    };
    // mean detrend in ONE pass over the signals (single chunk): the spectra stage subtracts an estimate mu0 of every channel's
    // mean and leaves its block sums; the exact means and B_i = sum_g X_i,g follow from them (k_op_finish per channel) and the
    // matrix is corrected at the end (in k_csdm_fold).  Saves the separate pass over all samples for the means
    const bool cm_onepass = detrend == 1 && !means_host && nframes <= mc && pipe_spec(nframes) && !env_flag("SP_CSDM_TWOPASS");
    if (cm_onepass) {
        HIPCHK(hipMemsetAsync(tb.f, 0, sizeof(float) * 4 * (size_t)nch, g.stream));   // (the spectra stage estimates and publishes mu0)
    } else if (means_host) {
        // caller-supplied constants (detrend == SP_DETREND_CONST semantics)
        double *md = tb.d;
        HIPCHK(hipMemcpyAsync(md, means_host, sizeof(double) * (size_t)nch, hipMemcpyHostToDevice, g.stream));
        hipLaunchKernelGGL(k_set_trends, dim3((nch + 63) / 64), dim3(64), 0, g.stream, md, tb.f, nch);
    } else if (detrend == 0) {
        HIPCHK(hipMemsetAsync(tb.f, 0, sizeof(float) * 4 * (size_t)nch, g.stream));
    } else if (nch <= 512) {
        double *scr = moments_scratch();                 // all channels in one launch
        if (!scr) return -1;
        LAUNCHCHK(launch_moments(lc(), xd, false, nsig, detrend, scr, tb.d, tb.f, nch, x_ld));
    } else {
        for (int c = 0; c < nch; ++c)
            if (set_trend(tb, c, xd + (size_t)x_ld * (size_t)c, false, nsig, detrend, 0, 0)) return -1;
    }
    size_t op_sp = 0, op_sl = 0;
    int op_runs = 0;
    bool fold_pending = false;                            // packed-spectra path used: H (g.cmH) is folded into G at the end
    for (int64_t f0 = 0; f0 < nframes; f0 += mc) {
        const int64_t m = nframes - f0 < mc ? nframes - f0 : mc;
        hipLaunchKernelGGL(k_trend_shift, dim3((nch + 63) / 64), dim3(64), 0, g.stream, tb.f, tb.f + 4 * nch, nch,
                           (double)f0 * (double)hop);
        if (lng) {
            // long segments: every channel's frames through pack -> batched long FFT, the rfft half of each row copied into Xs
            if (g.bigA.ensure(sizeof(cf) * (size_t)m * (size_t)nfft)) return -1;
            cf *S = (cf *)g.bigA.p;
            for (int c = 0; c < nch; ++c) {
                if (long_spectra(xd + (size_t)x_ld * (size_t)c, false, (const float *)win_d, nfft, hop, f0, m, tb.f + 4 * c,
                                 detrend == 2, 0, S, nullptr))
                    return -1;
                LAUNCHCHK(launch_long_stft_out(lc(), S, m, nfft, SP_SIDED_HALF, 1.f, 0, Xs + (size_t)c * (size_t)m * (size_t)ld, 0, ld));
            }
        } else if (pipe_spec(m)) {
            // nfft 4096 at 50 % overlap: the spectra stage is the pipeline of specialised waves (k_welch_pipe mode 5).  It writes
            // the PACKED pair spectra Z = X_2q + i X_2q+1 (all 4096 bins, two pairs side by side) and needs no mirror exchange;
            // the contraction runs on them as they are (H[k] = sum Z_i conj Z_j, same work: 4096 bins x pairs instead of 2049 x
            // frames) and the mirror combination G[k] = (H[k] + conj H[N-k]) / 2 is taken once at the end (k_csdm_fold)
            const int64_t pairs = (m + 1) / 2;
            const int runs = (g.ncu + nch - 1) / nch > 0 ? (g.ncu + nch - 1) / nch : 1;
            RunPart rp;
            rp.fpg = ((pairs + runs - 1) / runs + 1) & ~(int64_t)1;               // even: every run starts at an even pair
            rp.blocks = (int)((pairs + rp.fpg - 1) / rp.fpg);
            rp.groups = rp.blocks;
            const size_t zbytes = (size_t)128 * 64 * 512 * (size_t)((pairs + 1) / 2);
            const size_t hbytes = sizeof(double) * 2 * (size_t)nfft * (size_t)nch * (size_t)nch;
            if (g.cmS.ensure(zbytes) || g.cmH.ensure(hbytes)) return -1;
            Xs = (cf *)g.cmS.p;
            const int h_init = fold_pending ? 0 : 1;             // first chunk: H is initialised by the contraction itself
            fold_pending = true;
            cf *spartial = nullptr;
            if (cm_onepass) {
                // [channel][run][hop] block sums, then per channel Sl (2 hop doubles) and the finish state (5 nfft + 8 doubles)
                op_sp = sizeof(cf) * (size_t)nch * (size_t)rp.blocks * (size_t)hop;
                op_sl = sizeof(double) * 2 * (size_t)hop * (size_t)nch;
                if (g.cmO.ensure(op_sp + op_sl + sizeof(double) * (size_t)(5 * nfft + 8) * (size_t)nch)) return -1;
                spartial = (cf *)g.cmO.p;
                op_runs = rp.blocks;
            }
            LAUNCHCHK(launch_welch_pipe(lc(), xd + (size_t)f0 * (size_t)hop, false, (const float *)win_d, hop, m, tb.f + 4 * nch, xf,
                                        (float *)Xs, rp, spartial, 5, nch, x_ld, nfft / 8));
            LAUNCHCHK(launch_csdm_bf16(lc(), Xs, Xt, nch, pairs, nfft, (double *)g.cmH.p, nfft, split2, h_init));
            continue;
        } else if (use_bf16) {
            const RunPart rp = run_partition_2d(xf.L, (m + 1) / 2, g.ncu, nch);
            LAUNCHCHK(launch_stft_rp(lc(), xd + (size_t)f0 * (size_t)hop, (const float *)win_d, hop, m, tb.f + 4 * nch,
                                     detrend == 2, xf, rp, SP_SIDED_HALF, 1.f, 2, Xs, nullptr, nch, x_ld,
                                     ((m + 1) / 2) * (int64_t)ld * 2, ld));
        } else if (rp_stft && (m >= 2 || ld != nb)) {
            // all channels in one grid, two real frames per transform
            // (channels x groups) workgroups: about 8 per CU in all, so that each amortises its twiddle prologue over a long
            // run of frame pairs (256 groups per channel = 16 pairs per workgroup paid ~9 % for it); SP_STFT_GPC=1: old rule
            const RunPart rp = env_flag("SP_STFT_GPC") ? run_partition(xf.L, (m + 1) / 2, g.ncu, 1)
                                                       : run_partition_2d(xf.L, (m + 1) / 2, g.ncu, nch);
            LAUNCHCHK(launch_stft_rp(lc(), xd + (size_t)f0 * (size_t)hop, (const float *)win_d, hop, m, tb.f + 4 * nch,
                                     detrend == 2, xf, rp, SP_SIDED_HALF, 1.f, 0, Xs, nullptr, nch, x_ld,
                                     (int64_t)m * ld, ld));
        } else {
            const RunPart rp = run_partition(xf.L, m, g.ncu, 2);
            for (int c = 0; c < nch; ++c)
                LAUNCHCHK(launch_stft(lc(), xd + (size_t)x_ld * (size_t)c + (size_t)f0 * (size_t)hop, false,
                                      (const float *)win_d, hop, m, tb.f + 4 * (nch + c), detrend == 2, xf, rp, SP_SIDED_HALF,
                                      1.f, 0, Xs + (size_t)c * (size_t)m * (size_t)nb, nullptr));
        }
        if (!g_zeroed) {
            HIPCHK(hipMemsetAsync(G, 0, gbytes, g.stream));
            g_zeroed = true;
        }
        if (use_bf16) {
            LAUNCHCHK(launch_csdm_bf16(lc(), Xs, Xt, nch, m, nb, G, ld, split2));
        } else if (use_fused) {
            LAUNCHCHK(launch_csdm_fused(lc(), Xs, Xt, nch, m, nb, G, ld));
        } else if (use_mfma) {
            const int64_t mp = (m + 31) / 32 * 32;
            LAUNCHCHK(launch_csdm_transpose_kgc(lc(), Xs, Xt, nch, nchp, m, mp, nb));   // (these two forms: ld == nb)
            LAUNCHCHK(launch_csdm_mfma(lc(), Xt, nch, nchp, mp, nb, G));
        } else {
            LAUNCHCHK(launch_csdm_transpose(lc(), Xs, Xt, nch, m, nb));
            LAUNCHCHK(launch_csdm_gemm(lc(), Xt, nch, m, nb, G));
        }
    }
    const double gscale = scale / (double)nframes;
    const int fold_init = (fold_pending && !g_zeroed) ? 1 : 0;          // nothing was added into G directly: the fold finishes G
    if (cm_onepass) {
        void *Wf_d;
        if (get_window_spectrum(win, nfft, xf, &Wf_d)) return -1;
        double *Sl = (double *)((char *)g.cmO.p + op_sp), *st = (double *)((char *)g.cmO.p + op_sp + op_sl);
        LAUNCHCHK(launch_cm_blocksums(lc(), (const cf *)g.cmO.p, nch, op_runs, hop, Sl));
        LAUNCHCHK(launch_op_finish_channels(lc(), xd, x_ld, nch, tb.f + 4 * nch, (const float *)win_d, Sl, (const cf *)Wf_d, hop,
                                            nframes, nsig, xf, st));
        LAUNCHCHK(launch_csdm_fold(lc(), (const double *)g.cmH.p, G, nch, nfft, st, (const cf *)Wf_d, tb.f + 4 * nch, nsig, nframes,
                                   gscale, fold_init));
    } else if (fold_pending) {
        LAUNCHCHK(launch_csdm_fold(lc(), (const double *)g.cmH.p, G, nch, nfft, nullptr, nullptr, nullptr, 0, 0, gscale, fold_init));
    }
    if (!fold_init) LAUNCHCHK(launch_csdm_finish(lc(), G, nch, nb, gscale, use_mfma ? 32 : SP_CM_B));
    if (!mem) {
        HIPCHK(hipMemcpyAsync(g_out, G, gbytes, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

int sp_csd_matrix(const float *x, int nch, int64_t nsig, int64_t x_ld, const float *win, int nfft, int hop,
                  int64_t nframes, int detrend, double scale, double *g_out, int mem) {
    return csd_matrix_impl("sp_csd_matrix", x, nch, nsig, x_ld, win, nfft, hop, nframes, detrend, nullptr, scale, g_out, mem);
}

int sp_csd_matrix_means(const float *x, int nch, int64_t nsig, int64_t x_ld, const float *win, int nfft, int hop,
                        int64_t nframes, const double *means, double scale, double *g_out, int mem) {
    if (!means) return fail("sp_csd_matrix_means: means is NULL");
    return csd_matrix_impl("sp_csd_matrix_means", x, nch, nsig, x_ld, win, nfft, hop, nframes, 0, means, scale, g_out, mem);
}

int sp_channel_means(const float *x, int nch, int64_t nsig, int64_t x_ld, double *means_out, int mem) {
    if (ensure_init()) return -1;
    if (nch < 1 || nch > 512 || nsig < 1 || x_ld < nsig) return fail("sp_channel_means: need 1 <= nch <= 512, 1 <= nsig <= x_ld");
    ApiLock lk;
    const float *xd = x;
    if (!mem) {
        const size_t ib = sizeof(float) * (size_t)x_ld * (size_t)nch;
        if (g.in0.ensure(ib)) return -1;
        HIPCHK(hipMemcpyAsync(g.in0.p, x, ib, hipMemcpyHostToDevice, g.stream));
        xd = (const float *)g.in0.p;
    }
    TrendBuf tb;
    if (get_trendbuf(nch, &tb)) return -1;
    double *scr = moments_scratch();
    if (!scr) return -1;
    LAUNCHCHK(launch_moments(lc(), xd, false, nsig, 1, scr, tb.d, tb.f, nch, x_ld));
    // tb.d holds 8 doubles per channel, the mean first
    HIPCHK(hipMemcpy2DAsync(means_out, sizeof(double), tb.d, sizeof(double) * 8, sizeof(double), (size_t)nch,
                            mem ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, g.stream));
    if (!mem) HIPCHK(hipStreamSynchronize(g.stream));
    return 0;
}

int sp_stft(const void *x, int x_dtype, int64_t nsig, const float *win, int nfft, int hop, int64_t nframes,
            int detrend, double mean_re, double mean_im, int sided, double amp_scale, int out_kind, int out_major,
            void *out, double *pseg_out, int mem) {
    if (ensure_init()) return -1;
    if (check_frames("sp_stft", nsig, nfft, hop, nframes)) return -1;
    if (sided < 1 || sided > 4) return fail("sp_stft: bad sided");
    if (detrend < 0 || detrend > 4) return fail("sp_stft: detrend must be 0..4");
    const int segmean = detrend == SP_DETREND_SEGMEAN ? 1 : (detrend == SP_DETREND_SEGLINEAR ? 2 : 0);   // per-window: generic kernel
    if (segmean) {
        detrend = SP_DETREND_CONST;
        mean_re = mean_im = 0.0;
    }
    ApiLock lk;
    const bool lng = !wg_capable(nfft);
    Xf xf;
    if (!lng && get_xf(nfft, &xf)) return -1;
    const bool cplx = x_dtype == SP_DTYPE_C64;
    const size_t esz = cplx ? 8 : 4;
    const size_t osz = out_kind ? 4 : 8;
    const size_t nb = (size_t)nbins_host(nfft, sided);
    const size_t obytes = osz * nb * (size_t)nframes;
    const void *xd = x;
    if (!mem) {
        if (g.in0.ensure(esz * (size_t)nsig)) return -1;
        HIPCHK(hipMemcpyAsync(g.in0.p, x, esz * (size_t)nsig, hipMemcpyHostToDevice, g.stream));
        xd = g.in0.p;
    }
    void *win_d;
    if (get_table(1, win, sizeof(float) * (size_t)nfft, &win_d, nullptr)) return -1;
    TrendBuf tb;
    if (get_trendbuf(1, &tb)) return -1;
    if (set_trend(tb, 0, xd, cplx, nsig, detrend, mean_re, mean_im)) return -1;
    void *fin = out;                     // final layout
    double *pseg_d = pseg_out;
    if (!mem) {
        if (g.out0.ensure(obytes + (pseg_out ? sizeof(double) * (size_t)nframes + 16 : 0))) return -1;
        fin = g.out0.p;
        if (pseg_out) pseg_d = (double *)((char *)g.out0.p + ((obytes + 15) & ~(size_t)15));
    }
    void *fm = fin;                      // frame-major result; bin-major needs a transpose into `fin`
    if (out_major == 1) {
        if (g.work.ensure(obytes)) return -1;
        fm = g.work.p;
    }
    if (pseg_d) HIPCHK(hipMemsetAsync(pseg_d, 0, sizeof(double) * (size_t)nframes, g.stream));
    const bool pair = !lng && !cplx && !segmean && !xf.blue && xf.L >= 32 && nframes >= 2 && !env_flag("SP_NO_REALPAIR");
    if (lng) {
        const int64_t mc = long_chunk_frames(nfft, nframes);
        if (g.bigA.ensure(sizeof(cf) * (size_t)mc * (size_t)nfft)) return -1;
        cf *S = (cf *)g.bigA.p;
        for (int64_t f0 = 0; f0 < nframes; f0 += mc) {
            const int64_t m = nframes - f0 < mc ? nframes - f0 : mc;
            if (long_spectra(xd, cplx, (const float *)win_d, nfft, hop, f0, m, tb.f, detrend == 2, segmean, S, pseg_d)) return -1;
            LAUNCHCHK(launch_long_stft_out(lc(), S, m, nfft, sided, (float)amp_scale, out_kind, fm, f0, (int)nb));
        }
    } else if (pair) {
        const RunPart rp = run_partition(xf.L, (nframes + 1) / 2, g.ncu,
                                         stft_rp_groups_per_cu(xf, detrend == 2, hop, sided, out_kind, pseg_d != nullptr));
        LAUNCHCHK(launch_stft_rp(lc(), (const float *)xd, (const float *)win_d, hop, nframes, tb.f, detrend == 2, xf, rp, sided,
                                 (float)amp_scale, out_kind, fm, pseg_d));
    } else {
        const RunPart rp = run_partition(xf.L, nframes, g.ncu);
        LAUNCHCHK(launch_stft(lc(), xd, cplx, (const float *)win_d, hop, nframes, tb.f, detrend == 2, xf, rp, sided,
                              (float)amp_scale, out_kind, fm, pseg_d, segmean));
    }
    if (out_major == 1) LAUNCHCHK(launch_transpose(lc(), fm, fin, nframes, (int64_t)nb, (int)osz));
    if (!mem) {
        HIPCHK(hipMemcpyAsync(out, fin, obytes, hipMemcpyDeviceToHost, g.stream));
        if (pseg_out)
            HIPCHK(hipMemcpyAsync(pseg_out, pseg_d, sizeof(double) * (size_t)nframes, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

int sp_stft_cog(const void *x, int x_dtype, int64_t nsig, const float *win, int nfft, int hop, int64_t nframes,
                int detrend, double mean_re, double mean_im, double fs, double fmin, double fmax, double *cog_out, int mem) {
    if (ensure_init()) return -1;
    if (check_frames("sp_stft_cog", nsig, nfft, hop, nframes)) return -1;
    if (detrend < 0 || detrend > 4) return fail("sp_stft_cog: detrend must be 0..4");
    if (!(fs > 0.0) || fmin < 0.0 || fmax < fmin) return fail("sp_stft_cog: need fs > 0 and 0 <= fmin <= fmax");
    const int segmean = detrend == SP_DETREND_SEGMEAN ? 1 : (detrend == SP_DETREND_SEGLINEAR ? 2 : 0);
    if (segmean) {
        detrend = SP_DETREND_CONST;
        mean_re = mean_im = 0.0;
    }
    ApiLock lk;
    const bool lng = !wg_capable(nfft);
    Xf xf;
    if (!lng && get_xf(nfft, &xf)) return -1;
    const bool cplx = x_dtype == SP_DTYPE_C64;
    const size_t esz = cplx ? 8 : 4;
    const void *xd = x;
    if (!mem) {
        if (g.in0.ensure(esz * (size_t)nsig)) return -1;
        HIPCHK(hipMemcpyAsync(g.in0.p, x, esz * (size_t)nsig, hipMemcpyHostToDevice, g.stream));
        xd = g.in0.p;
    }
    void *win_d;
    if (get_table(1, win, sizeof(float) * (size_t)nfft, &win_d, nullptr)) return -1;
    TrendBuf tb;
    if (get_trendbuf(1, &tb)) return -1;
    // band in bins of df = fs/nfft: klo = ceil(fmin/df), khi = floor(fmax/df), with a relative guard against a frequency
    // that is a bin centre up to rounding
    const double df = fs / (double)nfft;
    const double eps = 1e-9;
    const int klo = (int)std::ceil(fmin / df - eps);
    const double kh = std::floor(fmax / df + eps);
    const int khi = kh > (double)nfft ? nfft : (int)kh;
    const int wpf = lng ? 1 : (xf.L / 16 >= 64 ? xf.L / 16 / 64 : 1);   // waves per frame: one (num, den) slot each
    // mean detrend in ONE pass (pipeline mode 8) when the window's spectrum is confined to a few bins: per-frame results cannot be
    // corrected afterwards in general, but |X - d W|^2 differs from |X|^2 only where W is non-zero -- 3 bins for Hann.  The
    // kernel keeps those bins of every frame and k_cog_finish_op corrects the moments with the exact mean.  SP_COG_TWOPASS=1:
    // the separate pass for the mean (0.35 of 0.79 ms at 2^28 complex64 samples)
    const bool cog_pipe = !lng && !segmean && detrend != 2 && klo <= 0 && khi >= nfft / 2 && !env_flag("SP_COG_GENERIC") &&
                          welch_pipe_wanted(xf, hop, nframes);
    CogLobe lobe_w{};
    const bool cog_op = cog_pipe && detrend == SP_DETREND_MEAN && xf.L == 4096 && (hop == 2048 || hop == 1024) &&
                        !env_flag("SP_COG_TWOPASS") && cog_window_lobe(win, nfft, &lobe_w);
    if (cog_op) {
        HIPCHK(hipMemsetAsync(tb.f, 0, sizeof(float) * 4, g.stream));            // (the kernel publishes its estimate here)
    } else if (set_trend(tb, 0, xd, cplx, nsig, detrend, mean_re, mean_im)) {
        return -1;
    }
    const size_t abytes = sizeof(cf) * (size_t)wpf * (size_t)nframes + (cog_op ? sizeof(cf) * 8 * (size_t)nframes : 0);
    if (g.work.ensure(abytes)) return -1;
    cf *acc = (cf *)g.work.p;
    double *fin = cog_out;
    if (!mem) {
        if (g.out0.ensure(sizeof(double) * (size_t)nframes)) return -1;
        fin = (double *)g.out0.p;
    }
    if (lng) {
        const int64_t mc = long_chunk_frames(nfft, nframes);
        if (g.bigA.ensure(sizeof(cf) * (size_t)mc * (size_t)nfft)) return -1;
        cf *S = (cf *)g.bigA.p;
        for (int64_t f0 = 0; f0 < nframes; f0 += mc) {
            const int64_t m = nframes - f0 < mc ? nframes - f0 : mc;
            if (long_spectra(xd, cplx, (const float *)win_d, nfft, hop, f0, m, tb.f, detrend == 2, segmean, S, nullptr)) return -1;
            LAUNCHCHK(launch_long_cog(lc(), S, m, nfft, klo, khi, acc, f0));
        }
    } else {
        const bool pipe = cog_pipe;
        const RunPart rp = pipe ? run_partition(xf.L, nframes, g.ncu, welch_pipe_gpc()) : run_partition(xf.L, nframes, g.ncu);
        // streaming form (every sample read once, the overlap carried in registers) when the shape allows; SP_COG_GENERIC=1
        // forces the generic frame kernel (A/B test); nfft 4096: the pipeline of specialised waves (k_welch_pipe.hip, mode 2)
        int generic = 1;
        if (cog_op) {
            // (Sl: 2 hop doubles; k_op_finish also reads nfft doubles through its unused A argument, hence the larger of the two)
            const size_t b_sp = sizeof(cf) * (size_t)rp.groups * (size_t)hop, b_sl = sizeof(double) * (size_t)(2 * hop > nfft ? 2 * hop : nfft);
            if (g.cmO.ensure(b_sp + b_sl + sizeof(double) * (size_t)(5 * nfft + 8))) return -1;
            cf *spartial = (cf *)g.cmO.p;
            double *Sl = (double *)((char *)g.cmO.p + b_sp), *st = (double *)((char *)g.cmO.p + b_sp + b_sl);
            void *Wf_d;
            if (get_window_spectrum(win, nfft, xf, &Wf_d)) return -1;
            LAUNCHCHK(launch_welch_pipe(lc(), xd, cplx, (const float *)win_d, hop, nframes, tb.f, xf, (float *)acc, rp, spartial, 2));
            LAUNCHCHK(launch_cm_blocksums(lc(), spartial, 1, (int)rp.groups, hop, Sl));
            LAUNCHCHK(launch_op_finish_channels(lc(), xd, 0, 1, tb.f, (const float *)win_d, Sl, (const cf *)Wf_d, hop, nframes, nsig, xf,
                                                st, cplx));
            LAUNCHCHK(launch_cog_finish_op(lc(), acc, wpf, nframes, df, fin, acc + (size_t)wpf * (size_t)nframes, lobe_w, st, tb.f, nsig,
                                           nfft));
            if (!mem) {
                HIPCHK(hipMemcpyAsync(cog_out, fin, sizeof(double) * (size_t)nframes, hipMemcpyDeviceToHost, g.stream));
                HIPCHK(hipStreamSynchronize(g.stream));
            }
            return 0;
        }
        if (pipe) {
            LAUNCHCHK(launch_welch_pipe(lc(), xd, cplx, (const float *)win_d, hop, nframes, tb.f, xf, (float *)acc, rp, nullptr, 2));
            generic = 0;
        } else if (!segmean && detrend != 2 && !env_flag("SP_COG_GENERIC"))
            generic = launch_cog_carry(lc(), xd, cplx, (const float *)win_d, hop, nframes, tb.f, xf, acc, rp, klo, khi);
        if (generic < 0) return fail("sp_stft_cog: launch failed");
        if (generic)
            LAUNCHCHK(launch_stft(lc(), xd, cplx, (const float *)win_d, hop, nframes, tb.f, detrend == 2, xf, rp, 2, 1.f, 1, nullptr,
                                  nullptr, segmean, acc, klo, khi));
    }
    LAUNCHCHK(launch_cog_finish(lc(), acc, wpf, nframes, df, fin));
    if (!mem) {
        HIPCHK(hipMemcpyAsync(cog_out, fin, sizeof(double) * (size_t)nframes, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

int sp_hilbert(const float *x, int64_t n_in, int64_t x_ld, int64_t nfft, int64_t batch, void *out, int mem) {
    if (ensure_init()) return -1;
    if (n_in < 1 || nfft < 2 || batch < 1 || x_ld < n_in) return fail("sp_hilbert: bad sizes");
    ApiLock lk;
    const float *xd = x;
    cf *od = (cf *)out;
    const size_t ibytes = sizeof(float) * (size_t)x_ld * (size_t)batch;
    const size_t obytes = sizeof(cf) * (size_t)nfft * (size_t)batch;
    if (!mem) {
        if (g.in0.ensure(ibytes) || g.out0.ensure(obytes)) return -1;
        HIPCHK(hipMemcpyAsync(g.in0.p, x, ibytes, hipMemcpyHostToDevice, g.stream));
        xd = (const float *)g.in0.p;
        od = (cf *)g.out0.p;
    }
    const int64_t nuse = n_in < nfft ? n_in : nfft;
    if (wg_capable(nfft)) {
        Xf xf;
        if (get_xf(nfft, &xf)) return -1;
        LAUNCHCHK(launch_hilbert(lc(), xd, nuse, x_ld, batch, xf, od));     // fwd + mask + inverse fused, one workgroup per row
    } else {
        // long rows: pack -> FFT -> mask -> inverse FFT, each a multi-pass transform
        if (g.bigA.ensure(sizeof(cf) * (size_t)nfft)) return -1;
        cf *A = (cf *)g.bigA.p;
        const int64_t Mh = nfft / 2;
        const bool half_len = (nfft & (nfft - 1)) == 0 && big_three_pass(Mh) && !env_flag("SP_LONG_NOFUSE") &&
                              !env_flag("SP_HILBERT_FULL");
        BigTw btN;
        if (half_len && get_bigtw(nfft, &btN)) return -1;
        for (int64_t b = 0; b < batch; ++b) {
            if (half_len && (((uintptr_t)(od + b * nfft)) & 15) == 0) {
                // real input: ONE half-length transform each way (z[n] = x[2n] + i x[2n+1], M = nfft/2 points).  Forward with
                // the pair load fused into its first pass; k_hilbert_mid turns Z into the half-length spectrum of the
                // Hilbert transform y (real-FFT split, analytic mask and inverse merge in one in-place step); the inverse
                // transform's last pass writes a = x + i y.  960 MB of traffic at 2^24 points instead of 1.45 GB, on 64 MB
                // buffers.
                BigFuse fz;
                const float *row = xd + b * x_ld;
                fz.ci = ColsIn{3, row, (((uintptr_t)row) & 7) == 0 ? row : nullptr, nullptr, nuse};      // r2 == r1: aligned pairs
                if (!env_flag("SP_HILBERT_NOFUSEMID")) {
                    // round 3: the middle step inside a row pass that owns mirror row pairs, the inverse as the adjoint passes in
                    // reversed order (k_hilbert_rowsmid, k_fft_cols_inv): 5 launches, 768 MB at 2^24 samples instead of 7 / 960
                    int lg = 0;
                    while (((int64_t)1 << lg) < Mh) ++lg;
                    int la_ = (lg + 2) / 3;
                    if (la_ > 8) la_ = 8;
                    int lb_ = (lg - la_ + 1) / 2;
                    if (lb_ > 8) lb_ = 8;
                    const int lc_ = lg - la_ - lb_;
                    const int64_t Aa = (int64_t)1 << la_, Bb = (int64_t)1 << lb_, Cc = (int64_t)1 << lc_;
                    Xf xa, xb, xc;
                    BigTw btM;
                    if (get_xf(Aa, &xa) || get_xf(Bb, &xb) || get_xf(Cc, &xc) || get_bigtw(Mh, &btM)) return -1;
                    const cf *tw2c = nullptr;                     // exp(-2 pi i m / (2 C)): the row pass's share of W_N^k
                    if (get_twiddles(2 * Cc, &tw2c)) return -1;
                    if (Aa >= 64 && Bb >= 64 && Cc >= 32 && Cc <= 2048) {
                        // (a full, 8-byte aligned row IS the complex sequence z: the plain first pass at three workgroups per CU
                        //  instead of the predicated pair load at two -- 58 -> 38 us at 2^24 samples)
                        if (nuse == nfft && fz.ci.r2 != nullptr && !env_flag("SP_HILBERT_PAIRLOAD")) {
                            LAUNCHCHK(launch_fft_cols(lc(), reinterpret_cast<const cf *>(row), A, Bb * Cc, 1, Bb * Cc, 0, 1, 0, xa, btM));
                        } else {
                            LAUNCHCHK(launch_fft_cols(lc(), A, A, Bb * Cc, 1, Bb * Cc, 0, 1, 0, xa, btM, 0, fz.ci));
                        }
                        LAUNCHCHK(launch_fft_cols(lc(), A, A, Cc, Aa, Cc, Bb * Cc, Aa, 0, xb, btM));
                        LAUNCHCHK(launch_hilbert_rowsmid(lc(), A, Aa, Bb, xc, btN, tw2c));
                        LAUNCHCHK(launch_fft_cols_inv(lc(), A, A, Cc, Aa, Cc, Bb * Cc, Aa, xb, btM, 1.f, nullptr));
                        RowsOut ao;
                        ao.co = reinterpret_cast<float *>(od + b * nfft);
                        ao.n = nuse;
                        ao.Ltot = nfft;
                        ao.mom = nullptr;
                        ao.rx = row;
                        ao.kind = 2;
                        LAUNCHCHK(launch_fft_cols_inv(lc(), A, nullptr, Bb * Cc, 1, Bb * Cc, 0, 1, xa, btM, (float)(1.0 / (double)Mh), &ao));
                        continue;
                    }
                }
                if (dev_fft_big_pow2(A, A, Mh, 0, 0, 1, &fz)) return -1;
                LAUNCHCHK(launch_hilbert_mid(lc(), A, Mh, btN));
                BigFuse fo;
                fo.ro.co = reinterpret_cast<float *>(od + b * nfft);
                fo.ro.n = nuse;
                fo.ro.Ltot = nfft;
                fo.ro.rx = xd + b * x_ld;
                fo.ro.kind = 2;
                if (dev_fft_big_pow2(A, A, Mh, 1, 0, 1, &fo)) return -1;
                continue;
            }
            if (big_three_pass(nfft) && !env_flag("SP_LONG_NOFUSE")) {
                // real -> complex pack fused into the first pass of the forward transform (the zero padding is not loaded),
                // the analytic-signal mask into the first pass of the inverse one
                BigFuse fz;
                fz.ci = ColsIn{1, xd + b * x_ld, nullptr, nullptr, nuse};
                if (dev_fft_big_pow2(A, A, nfft, 0, 0, 1, &fz)) return -1;
                if (dev_fft_big_pow2(A, od + b * nfft, nfft, 1, 1)) return -1;
                continue;
            }
            LAUNCHCHK(launch_pack_real(lc(), xd + b * x_ld, nuse, nullptr, nfft, A));
            if (dev_fft_any(A, A, nfft, 1, 0)) return -1;
            if (big_three_pass(nfft)) {
                // the mask rides on the first pass of the inverse transform
                if (dev_fft_big_pow2(A, od + b * nfft, nfft, 1, 1)) return -1;
            } else {
                LAUNCHCHK(launch_hilbert_mask(lc(), A, nfft));
                if (dev_fft_any(A, od + b * nfft, nfft, 1, 1)) return -1;
            }
        }
    }
    if (!mem) {
        HIPCHK(hipMemcpyAsync(out, od, obytes, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

int sp_frame_sum(const void *x, int x_dtype, int64_t nsig, int nch, int64_t x_ld, int nfft, int hop, int64_t nframes,
                 int detrend, double *out, int mem) {
    if (ensure_init()) return -1;
    if (check_frames("sp_frame_sum", nsig, nfft, hop, nframes)) return -1;
    if (nch < 1 || nch > 65535 || x_ld < nsig) return fail("sp_frame_sum: bad nch / x_ld");
    if (detrend < 0 || detrend > 2) return fail("sp_frame_sum: detrend must be 0, 1 or 2");
    ApiLock lk;
    const bool cplx = x_dtype == SP_DTYPE_C64;
    const size_t esz = cplx ? 8 : 4;
    const void *xd = x;
    const size_t ibytes = esz * ((size_t)x_ld * (size_t)(nch - 1) + (size_t)nsig);
    const size_t obytes = sizeof(double) * 2 * (size_t)nfft * (size_t)nch;
    if (!mem) {
        if (g.in1.ensure(ibytes)) return -1;
        HIPCHK(hipMemcpyAsync(g.in1.p, x, ibytes, hipMemcpyHostToDevice, g.stream));
        xd = g.in1.p;
    }
    TrendBuf tb;
    if (get_trendbuf(nch + 1, &tb)) return -1;
    if (detrend != 0 && nch <= 512) {
        double *scr = moments_scratch();
        if (!scr) return -1;
        LAUNCHCHK(launch_moments(lc(), xd, cplx, nsig, detrend, scr, tb.d + 8, tb.f + 4, nch, x_ld));
    } else {
        for (int c = 0; c < nch; ++c)
            if (set_trend(tb, c + 1, (const char *)xd + esz * (size_t)x_ld * (size_t)c, cplx, nsig, detrend, 0, 0)) return -1;
    }
    double *od = out;
    if (!mem) {
        if (g.out0.ensure(obytes)) return -1;
        od = (double *)g.out0.p;
    }
    const size_t pbytes = obytes * (size_t)frame_sum_slices(g.ncu, nch, nfft, nframes);
    if (g.work.ensure(pbytes)) return -1;
    LAUNCHCHK(launch_frame_sum(lc(), xd, cplx, x_ld, nch, nfft, hop, nframes, tb.f + 4, detrend == 2, od, (double *)g.work.p));
    if (!mem) {
        HIPCHK(hipMemcpyAsync(out, od, obytes, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

int sp_spectral_filter(const float *x, int64_t n_in, int64_t x_ld, int64_t nfft, int64_t batch, const void *H, void *out,
                       int mem) {
    if (ensure_init()) return -1;
    if (n_in < 1 || nfft < 2 || batch < 1 || x_ld < n_in || H == nullptr) return fail("sp_spectral_filter: bad sizes");
    ApiLock lk;
    const float *xd = x;
    cf *od = (cf *)out;
    const size_t ibytes = sizeof(float) * (size_t)x_ld * (size_t)batch;
    const size_t obytes = sizeof(cf) * (size_t)nfft * (size_t)batch;
    const size_t hbytes = sizeof(cf) * (size_t)nfft;
    if (!mem) {
        if (g.in0.ensure(ibytes) || g.out0.ensure(obytes)) return -1;
        HIPCHK(hipMemcpyAsync(g.in0.p, x, ibytes, hipMemcpyHostToDevice, g.stream));
        xd = (const float *)g.in0.p;
        od = (cf *)g.out0.p;
    }
    const int64_t nuse = n_in < nfft ? n_in : nfft;
    if (wg_capable(nfft)) {
        Xf xf;
        if (get_xf(nfft, &xf)) return -1;
        void *H_d;
        if (get_table(4, H, hbytes, &H_d, nullptr)) return -1;          // the response table is a host array (like win)
        LAUNCHCHK(launch_hilbert(lc(), xd, nuse, x_ld, batch, xf, od, (const cf *)H_d));   // fwd, x H, inverse: one workgroup per row
    } else {
        if (g.bigA.ensure(sizeof(cf) * (size_t)nfft) || g.bigB.ensure(hbytes)) return -1;
        cf *A = (cf *)g.bigA.p, *Hd = (cf *)g.bigB.p;
        HIPCHK(hipMemcpyAsync(Hd, H, hbytes, hipMemcpyHostToDevice, g.stream));
        for (int64_t b = 0; b < batch; ++b) {
            LAUNCHCHK(launch_pack_real(lc(), xd + b * x_ld, nuse, nullptr, nfft, A));
            if (dev_fft_any(A, A, nfft, 1, 0)) return -1;
            LAUNCHCHK(launch_spec_mul(lc(), A, Hd, nfft));
            if (dev_fft_any(A, od + b * nfft, nfft, 1, 1)) return -1;
        }
        HIPCHK(hipStreamSynchronize(g.stream));        // H was read from the caller's host array
    }
    if (!mem) {
        HIPCHK(hipMemcpyAsync(out, od, obytes, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

int sp_xcorr(const float *x1, const float *x2, int64_t n, float *co_out, int mem) {
    if (ensure_init()) return -1;
    if (n < 1) return fail("sp_xcorr: n must be positive");
    const int64_t L = next_pow2(2 * n) < 2 ? 2 : next_pow2(2 * n);
    if (L > ((int64_t)1 << SP_MAX_BIG_LOG2))
        return fail("sp_xcorr: n=%lld needs a %lld-point transform; the limit is 2^%d", (long long)n, (long long)L, SP_MAX_BIG_LOG2);
    ApiLock lk;
    const float *a = x1, *b = x2;
    float *od = co_out;
    const size_t ibytes = sizeof(float) * (size_t)n, obytes = sizeof(float) * (size_t)(2 * n - 1);
    if (!mem) {
        if (g.in0.ensure(ibytes) || g.in1.ensure(ibytes) || g.out0.ensure(obytes)) return -1;
        HIPCHK(hipMemcpyAsync(g.in0.p, x1, ibytes, hipMemcpyHostToDevice, g.stream));
        HIPCHK(hipMemcpyAsync(g.in1.p, x2, ibytes, hipMemcpyHostToDevice, g.stream));
        a = (const float *)g.in0.p;
        b = (const float *)g.in1.p;
        od = (float *)g.out0.p;
    }
    TrendBuf tb;
    if (get_trendbuf(3, &tb)) return -1;
    double *scr = moments_scratch();
    if (!scr) return -1;
    // both signals' mean / variance in ONE pair of launches (two "channels" b - a samples apart: any two device rows; each
    // signal's own pass cost 21 + 8 us of the long ccf's 0.9 ms)
    // (and the normalisation record behind them: one finish launch for both signals, no k_xcorr_norm)
    LAUNCHCHK(launch_moments_xc(lc(), a, n, scr, tb.d, tb.d + 16, (int64_t)(b - a)));
    if (L <= SP_MAX_WG_FFT) {
        Xf xf;
        if (get_xf(L, &xf)) return -1;
        LAUNCHCHK(launch_xcorr(lc(), a, b, n, tb.d + 16, xf, od));          // one workgroup, everything in registers/LDS
    } else {
        if (g.bigA.ensure(sizeof(cf) * (size_t)L) || g.bigB.ensure(sizeof(cf) * (size_t)L)) return -1;
        cf *A = (cf *)g.bigA.p, *B = (cf *)g.bigB.p;
        if (big_three_pass(L) && !env_flag("SP_LONG_NOFUSE")) {
            // two of the three elementwise kernels ride on the transforms: the pack in the first pass of the first transform
            // (the zero half is never loaded), lag re-ordering + real part in the last pass of the second
            BigFuse f1, f2;
            f1.ci = ColsIn{1, a, b, tb.d + 16, n};
            if (big_three_pass(L / 2) && !env_flag("SP_XC_FULL") && !env_flag("SP_XC_NOFUSEMID")) {
                // round 3: the forward transform's row pass, the middle step and the FIRST pass of the half-length transform in one
                // kernel that owns mirror row pairs (k_xc_rowsmid); the half-length transform's two remaining passes are column
                // passes over the half rows it leaves ([ka][kb][ka'], C/2 per row), the last one writing the lags
                int lg = 0;
                while (((int64_t)1 << lg) < L) ++lg;
                int la_ = (lg + 2) / 3;
                if (la_ > 8) la_ = 8;
                int lb_ = (lg - la_ + 1) / 2;
                if (lb_ > 8) lb_ = 8;
                const int lc_ = lg - la_ - lb_;
                const int64_t Aa = (int64_t)1 << la_, Bb = (int64_t)1 << lb_, Cc = (int64_t)1 << lc_, C2 = Cc / 2;
                Xf xa, xb, xc, xc2;
                BigTw btL, btM;
                if (get_xf(Aa, &xa) || get_xf(Bb, &xb) || get_xf(Cc, &xc) || get_xf(C2, &xc2) || get_bigtw(L, &btL) ||
                    get_bigtw(L / 2, &btM))
                    return -1;
                if (Aa >= 64 && Aa <= 256 && Bb >= 64 && Bb <= 256 && Cc >= 64 && Cc <= 2048) {
                    LAUNCHCHK(launch_fft_cols(lc(), A, A, Bb * Cc, 1, Bb * Cc, 0, 1, 0, xa, btL, 0, f1.ci));
                    LAUNCHCHK(launch_fft_cols(lc(), A, A, Cc, Aa, Cc, Bb * Cc, Aa, 0, xb, btL));
                    LAUNCHCHK(launch_xc_rowsmid(lc(), A, Aa, Bb, xc, xc2, btL, btM));
                    // M-point transform, split (A' = C/2 [done], B' = B, C' = A) on the layout [ka][kb][ka']:
                    //   over kb (stride = the row pitch C, columns ka' < C/2, outer ka), twiddle W_M^{A' ka kb''}: by the OUTER index
                    LAUNCHCHK(launch_fft_cols(lc(), A, A, C2, Aa, Cc, Bb * Cc, C2, 0, xb, btM, 0, ColsIn{0, nullptr, nullptr, nullptr, 0}, 1));
                    //   over ka (stride B C): natural index k' = ka' + A' kb'' + A' B' kc''; the lags leave from here
                    RowsOut ro{od, n, L, tb.d + 16};
                    ro.kind = 3;
                    LAUNCHCHK(launch_fft_cols_lag(lc(), A, C2, Bb, Bb * Cc, Cc, xa, ro));
                    if (!mem) {
                        HIPCHK(hipMemcpyAsync(co_out, od, obytes, hipMemcpyDeviceToHost, g.stream));
                        HIPCHK(hipStreamSynchronize(g.stream));
                    }
                    return 0;
                }
            }
            if (dev_fft_big_pow2(A, B, L, 0, 0, 1, &f1)) return -1;          // B = FFT(z)
            if (big_three_pass(L / 2) && !env_flag("SP_XC_FULL")) {
                // the correlation is real: its inverse transform runs at half length (k_xc_mid_half forms the M-point spectrum
                // of r[2n] + i r[2n+1]; the last pass writes two lags per element): 3.0 GB of traffic at 2^24 samples
                // instead of 3.8
                BigTw btL;
                if (get_bigtw(L, &btL)) return -1;
                LAUNCHCHK(launch_xc_mid_half(lc(), B, L, btL, A));
                f2.ro = RowsOut{od, n, L, tb.d + 16};
                f2.ro.kind = 3;
                if (dev_fft_big_pow2(A, A, L / 2, 0, 0, 1, &f2)) return -1;
            } else {
                LAUNCHCHK(launch_xc_mid(lc(), B, L, A));                     // conj(A conj(B)) spectrum
                f2.ro = RowsOut{od, n, L, tb.d + 16};
                f2.ro.kind = 1;
                if (dev_fft_big_pow2(A, B, L, 0, 0, 1, &f2)) return -1;      // writes co from its last pass
            }
        } else {
            LAUNCHCHK(launch_xc_pack(lc(), a, b, n, L, tb.d + 16, A));
            if (dev_fft_big_pow2(A, B, L, 0)) return -1;
            LAUNCHCHK(launch_xc_mid(lc(), B, L, A));                         // conj(A conj(B)) spectrum
            if (dev_fft_big_pow2(A, B, L, 0)) return -1;                     // forward of the conjugate = L * real inverse
            LAUNCHCHK(launch_xc_out(lc(), B, n, L, tb.d + 16, od));
        }
    }
    if (!mem) {
        HIPCHK(hipMemcpyAsync(co_out, od, obytes, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

int sp_csd_epilogue(const double *pxx, const double *pyy, const double *pxy, int nch, int nb, int nfft, int onesided, double enbw,
                    double *out, int mem) {
    if (ensure_init()) return -1;
    if (nch < 1 || nch > 20000 || nb < 1 || nfft < 2 || nb > nfft) return fail("sp_csd_epilogue: bad sizes");
    if (onesided && nb != nbins_host(nfft, SP_SIDED_ONE)) return fail("sp_csd_epilogue: one-sided spectra hold %d bins for nfft=%d", nbins_host(nfft, SP_SIDED_ONE), nfft);
    if (!onesided && nb != nfft) return fail("sp_csd_epilogue: two-sided spectra hold nfft bins");
    ApiLock lk;
    const size_t NB = (size_t)nb, NF = (size_t)nfft, C = (size_t)nch;
    const size_t n_in = NB * (1 + 3 * C);                                  // pxx | pyy | pxy (complex)
    const size_t n_out = sp_csd_epilogue_doubles(nch, nb, nfft);
    const double *dxx = pxx, *dyy = pyy, *dxy = pxy;
    double *od = out;
    if (!mem) {
        if (g.out0.ensure(sizeof(double) * (n_in + n_out))) return -1;
        double *st = (double *)g.out0.p;
        HIPCHK(hipMemcpyAsync(st, pxx, sizeof(double) * NB, hipMemcpyHostToDevice, g.stream));
        HIPCHK(hipMemcpyAsync(st + NB, pyy, sizeof(double) * NB * C, hipMemcpyHostToDevice, g.stream));
        HIPCHK(hipMemcpyAsync(st + NB * (1 + C), pxy, sizeof(double) * 2 * NB * C, hipMemcpyHostToDevice, g.stream));
        dxx = st;
        dyy = st + NB;
        dxy = st + NB * (1 + C);
        od = st + n_in;
    }
    // out layout (doubles): cxy[C][NB][2] | cxy2[C][NB] | phi[C][NB] | lxx[NB] | lyy[C][NB] | lxy[C][NB] |
    //                       rxx[NF][2] | ryy[C][NF][2] | rxy[C][NF][2] | icxy[C][NF][2] | corrcoef[C][NF][2] | e[1+C][2]
    double *cxy = od, *cxy2 = cxy + 2 * C * NB, *phi = cxy2 + C * NB, *lxx = phi + C * NB, *lyy = lxx + NB, *lxy = lyy + C * NB;
    double *rxx = lxy + C * NB, *ryy = rxx + 2 * NF, *rxy = ryy + 2 * C * NF, *icxy = rxy + 2 * C * NF, *cc = icxy + 2 * C * NF;
    double *ee = cc + 2 * C * NF;
    LAUNCHCHK(launch_epi_elem(lc(), dxx, dyy, dxy, nch, nb, nfft, onesided, enbw, cxy, cxy2, phi, lxx, lyy, lxy));
    const size_t nsig = 3 * C + 1;
    if (g.bigA.ensure(sizeof(cf) * nsig * NF) || g.epi.ensure(sizeof(double) * (C + 1))) return -1;
    cf *X = (cf *)g.bigA.p;
    // the rows' scales: the inverse transforms run in float32 on rows normalised to O(1) (k_epilogue.hip)
    double *rowmax = (double *)g.epi.p;
    LAUNCHCHK(launch_epi_spec(lc(), dxx, dyy, dxy, cxy, nch, nb, nfft, onesided, X, rowmax));
    if (dev_fft_any(X, X, nfft, (int64_t)nsig, 1)) return -1;
    LAUNCHCHK(launch_epi_corr(lc(), X, nch, nfft, onesided, rxx, ryy, rxy, icxy, ee, cc, rowmax));
    if (!mem) {
        HIPCHK(hipMemcpyAsync(out, od, sizeof(double) * n_out, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

int64_t sp_csd_epilogue_doubles(int nch, int nb, int nfft) {
    const int64_t C = nch, NB = nb, NF = nfft;
    return 2 * C * NB + C * NB + C * NB + NB + C * NB + C * NB + 2 * NF + 4 * 2 * C * NF + 2 * (1 + C);
}

int sp_biquad(const double *b, const double *a, const float *x, int64_t n, float *y, int mem) {
    if (ensure_init()) return -1;
    if (n < 1 || b == nullptr || a == nullptr) return fail("sp_biquad: bad arguments");
    if (a[0] == 0.0) return fail("sp_biquad: a[0] must not be zero");
    {   // the blocked scan raises the state map to powers up to n: an unstable section overflows them to inf and inf * 0 = NaN
        // would reach every output (scipy's lfilter returns the valid early samples) -- refuse instead
        const double a1 = a[1] / a[0], a2 = a[2] / a[0], disc = a1 * a1 - 4.0 * a2;
        const double rad = disc < 0.0 ? sqrt(a2) : 0.5 * (fabs(a1) + sqrt(disc));
        if (!(rad <= 1.0 + 1e-12))
            return fail("sp_biquad: unstable section (pole radius %.9g > 1): the blocked scan of the state maps does not apply", rad);
    }
    ApiLock lk;
    const float *xd = x;
    float *yd = y;
    const size_t bytes = sizeof(float) * (size_t)n;
    if (!mem) {
        if (g.in0.ensure(bytes) || g.out0.ensure(bytes)) return -1;
        HIPCHK(hipMemcpyAsync(g.in0.p, x, bytes, hipMemcpyHostToDevice, g.stream));
        xd = (const float *)g.in0.p;
        yd = (float *)g.out0.p;
    }
    if (g.work.ensure(sizeof(double) * 4 * (size_t)biquad_tiles(n) + 64)) return -1;
    LAUNCHCHK(launch_biquad(lc(), b, a, xd, n, yd, (double *)g.work.p));
    if (!mem) {
        HIPCHK(hipMemcpyAsync(y, yd, bytes, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

int sp_fftfilt(const float *h, int ntaps, const float *x, int64_t n, int nfft, float *y, int mem) {
    if (ensure_init()) return -1;
    if (ntaps < 1 || n < 1) return fail("sp_fftfilt: bad sizes");
    if (nfft == 0) {
        int64_t c = next_pow2(8 * (int64_t)ntaps);
        if (c < 1024) c = 1024;
        if (c > SP_MAX_WG_FFT) c = SP_MAX_WG_FFT;
        nfft = (int)c;
    }
    if (!(is_pow2(nfft) && nfft >= 2 && nfft <= SP_MAX_WG_FFT) || nfft < 2 * (ntaps - 1) || nfft <= ntaps - 1)
        return fail("sp_fftfilt: nfft=%d must be a power of two with 2*(ntaps-1) <= nfft <= %d", nfft, SP_MAX_WG_FFT);
    ApiLock lk;
    Xf xf;
    if (get_xf(nfft, &xf)) return -1;
    // Hs = FFT(h zero-padded)/nfft, cached per (taps, nfft)
    std::vector<cf> hp((size_t)nfft, make_float2(0.f, 0.f));
    for (int i = 0; i < ntaps; ++i) hp[(size_t)i] = make_float2(h[i] / (float)nfft, 0.f);
    void *H_d;
    bool fresh = false;
    if (get_table(2, hp.data(), sizeof(cf) * (size_t)nfft, &H_d, &fresh)) return -1;
    if (fresh) LAUNCHCHK(launch_fft_c2c(lc(), (const cf *)H_d, (cf *)H_d, 1, 0, xf));
    const float *xd = x;
    float *yd = y;
    const size_t bytes = sizeof(float) * (size_t)n;
    if (!mem) {
        if (g.in0.ensure(bytes) || g.out0.ensure(bytes)) return -1;
        HIPCHK(hipMemcpyAsync(g.in0.p, x, bytes, hipMemcpyHostToDevice, g.stream));
        xd = (const float *)g.in0.p;
        yd = (float *)g.out0.p;
    }
    LAUNCHCHK(launch_fftfilt(lc(), xd, n, ntaps, (const cf *)H_d, xf, yd));
    if (!mem) {
        HIPCHK(hipMemcpyAsync(y, yd, bytes, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

}   // extern "C"
