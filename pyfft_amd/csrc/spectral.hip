// spectral.hip -- C ABI (include/spectral.h) over the gfx950 kernels in kernels.h.
#include "../../include/spectral.h"
#include "kernels.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

using namespace sp;

#define SP_VERSION 100
#define SP_MAX_WG_FFT 8192

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
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

struct Ctx {
    bool ready = false;
    int device = 0;
    int ncu = 256;
    hipStream_t stream = nullptr;
    std::map<int64_t, cf *> twiddles;   // n -> device table exp(-2 pi i m/n), m < n
    Scratch in0, in1, out0, work, small;   // staging (mem=0) and workspace
    std::mutex mu;
    // optional timing of the dominant kernel of the last call (HIP events on the launch stream)
    bool profile = false;
    bool prof_valid = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
} g;

struct ProfScope {   // brackets one kernel launch with events when profiling is on
    bool on;
    ProfScope() : on(g.profile) {
        if (on) (void)hipEventRecord(g.ev0, g.stream);
    }
    ~ProfScope() {
        if (on) {
            (void)hipEventRecord(g.ev1, g.stream);
            g.prof_valid = true;
        }
    }
};

int ensure_init() {
    if (g.ready) return 0;
    return sp_init(-1);
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

bool is_pow2(int64_t n) { return n >= 1 && (n & (n - 1)) == 0; }

// number of transform groups to launch for a run-partitioned kernel: enough workgroups to fill the
// chip several times over, but at least `min_run` frames per group so the run amortises the twiddle /
// window prologue and the partial-spectrum store.
template <int N> void run_partition(int64_t nframes, int64_t *groups, int64_t *fpg, int *blocks) {
    using C = WgCfg<N>;
    const int64_t target_groups = (int64_t)g.ncu * 8 * C::FPW;
    int64_t f = (nframes + target_groups - 1) / target_groups;
    if (f < 1) f = 1;
    int64_t G = (nframes + f - 1) / f;
    int b = (int)((G + C::FPW - 1) / C::FPW);
    *fpg = f;
    *blocks = b;
    *groups = (int64_t)b * C::FPW;
}

template <int N> int strided_blocks(int64_t items) {
    using C = WgCfg<N>;
    int64_t b = (items + C::FPW - 1) / C::FPW;
    const int64_t cap = (int64_t)g.ncu * 16;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

#define SP_DISPATCH_N(n, MACRO)                                                                      \
    switch (n) {                                                                                     \
        case 2: MACRO(2); break;                                                                     \
        case 4: MACRO(4); break;                                                                     \
        case 8: MACRO(8); break;                                                                     \
        case 16: MACRO(16); break;                                                                   \
        case 32: MACRO(32); break;                                                                   \
        case 64: MACRO(64); break;                                                                   \
        case 128: MACRO(128); break;                                                                 \
        case 256: MACRO(256); break;                                                                 \
        case 512: MACRO(512); break;                                                                 \
        case 1024: MACRO(1024); break;                                                               \
        case 2048: MACRO(2048); break;                                                               \
        case 4096: MACRO(4096); break;                                                               \
        case 8192: MACRO(8192); break;                                                               \
        default: return fail("internal: no workgroup FFT for n=%lld", (long long)(n));               \
    }

// ---- device-side building blocks (all pointers device, enqueue on g.stream) --------------------

int dev_fft_pow2_wg(const cf *in, cf *out, int n, int64_t batch, int inverse) {
    const cf *tw;
    if (get_twiddles(n, &tw)) return -1;
#define L_(NN)                                                                                       \
    {                                                                                                \
        using C = WgCfg<NN>;                                                                         \
        const int blocks = strided_blocks<NN>(batch);                                                \
        hipLaunchKernelGGL((k_fft_c2c<NN>), dim3(blocks), dim3(C::WG), C::lds_bytes(1), g.stream, in, out, batch, \
                           inverse, tw);                                                             \
    }
    SP_DISPATCH_N(n, L_)
#undef L_
    HIPCHK(hipGetLastError());
    return 0;
}

int dev_moments(const void *x, int dtype, int64_t n, double *out_d /*[4] dev*/, float *out_f /*[2] dev or null*/) {
    const int threads = 256;
    int64_t nb = (n + threads * 8 - 1) / (threads * 8);
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    if (g.small.ensure(sizeof(double) * 4 * 2048 + 256)) return -1;
    double *partial = (double *)g.small.p;
    if (dtype == SP_DTYPE_C64)
        hipLaunchKernelGGL((k_moments_partial<true>), dim3((int)nb), dim3(threads), 0, g.stream, x, n, partial);
    else
        hipLaunchKernelGGL((k_moments_partial<false>), dim3((int)nb), dim3(threads), 0, g.stream, x, n, partial);
    hipLaunchKernelGGL(k_moments_finish, dim3(1), dim3(256), 0, g.stream, partial, (int)nb, n, out_d, out_f);
    HIPCHK(hipGetLastError());
    return 0;
}

// small device block of parameters: [0..1] float mean x, [2..] float mean y per channel; doubles after
struct MeanBuf {
    float *f = nullptr;    // device
    double *d = nullptr;   // device
};

Scratch g_means;

int get_meanbuf(int nch, MeanBuf *mb) {
    const size_t fbytes = sizeof(float) * 2 * (size_t)(nch + 1);
    const size_t fpad = (fbytes + 15) & ~(size_t)15;
    if (g_means.ensure(fpad + sizeof(double) * 4 * (size_t)(nch + 1))) return -1;
    mb->f = (float *)g_means.p;
    mb->d = (double *)((char *)g_means.p + fpad);
    return 0;
}

int dev_welch_psd(const void *x, int dtype, const float *win_d, int nfft, int hop, int64_t nframes,
                  const float *mean_d, int sided, double scale, double *out_d) {
    const cf *tw;
    if (get_twiddles(nfft, &tw)) return -1;
#define L_(NN)                                                                                       \
    {                                                                                                \
        using C = WgCfg<NN>;                                                                         \
        int64_t G, fpg;                                                                              \
        int blocks;                                                                                  \
        run_partition<NN>(nframes, &G, &fpg, &blocks);                                               \
        if (g.work.ensure(sizeof(float) * (size_t)G * NN)) return -1;                                \
        float *partial = (float *)g.work.p;                                                          \
        {                                                                                            \
            ProfScope ps_;                                                                           \
            if (dtype == SP_DTYPE_C64)                                                               \
                hipLaunchKernelGGL((k_welch<NN, true>), dim3(blocks), dim3(C::WG), C::lds_bytes(1), g.stream, x, \
                                   win_d, hop, nframes, fpg, mean_d, tw, partial);                   \
            else                                                                                     \
                hipLaunchKernelGGL((k_welch<NN, false>), dim3(blocks), dim3(C::WG), C::lds_bytes(1), g.stream, x, \
                                   win_d, hop, nframes, fpg, mean_d, tw, partial);                   \
        }                                                                                            \
        hipLaunchKernelGGL((k_welch_finish<NN>), dim3((NN + SP_FIN_BINS - 1) / SP_FIN_BINS),         \
                           dim3(SP_FIN_BINS * SP_FIN_SLICES), 0, g.stream, partial, G, sided,        \
                           scale / (double)nframes, out_d);                                          \
    }
    SP_DISPATCH_N(nfft, L_)
#undef L_
    HIPCHK(hipGetLastError());
    return 0;
}

int dev_welch_csd(const void *x, const void *y, int dtype, int nch, int64_t y_ld, const float *win_d, int nfft,
                  int hop, int64_t nframes, const float *mean_x_d, const float *mean_y_d, int sided, double scale,
                  double *pxx_d, double *pyy_d, double *pxy_d) {
    const cf *tw;
    if (get_twiddles(nfft, &tw)) return -1;
#define L_(NN)                                                                                       \
    {                                                                                                \
        using C = WgCfg<NN>;                                                                         \
        int64_t G, fpg;                                                                              \
        int blocks;                                                                                  \
        run_partition<NN>(nframes, &G, &fpg, &blocks);                                               \
        if (g.work.ensure(sizeof(float) * (size_t)G * NN * 4 * (size_t)nch)) return -1;              \
        float *partial = (float *)g.work.p;                                                          \
        if (dtype == SP_DTYPE_C64)                                                                   \
            hipLaunchKernelGGL((k_welch_csd<NN, true>), dim3(blocks, nch), dim3(C::WG), C::lds_bytes(1), g.stream, \
                               x, y, y_ld, win_d, hop, nframes, fpg, mean_x_d, mean_y_d, tw, partial, G); \
        else                                                                                         \
            hipLaunchKernelGGL((k_welch_csd<NN, false>), dim3(blocks, nch), dim3(C::WG), C::lds_bytes(1), g.stream, \
                               x, y, y_ld, win_d, hop, nframes, fpg, mean_x_d, mean_y_d, tw, partial, G); \
        hipLaunchKernelGGL((k_csd_finish<NN>), dim3((NN + SP_FIN_BINS - 1) / SP_FIN_BINS, nch),      \
                           dim3(SP_FIN_BINS * SP_FIN_SLICES), 0, g.stream, partial, G, nch, sided,   \
                           scale / (double)nframes, pxx_d, pyy_d, pxy_d);                            \
    }
    SP_DISPATCH_N(nfft, L_)
#undef L_
    HIPCHK(hipGetLastError());
    return 0;
}

int dev_stft(const void *x, int dtype, const float *win_d, int nfft, int hop, int64_t nframes, const float *mean_d,
             int sided, double amp, int out_kind, void *out_d, double *pseg_d) {
    const cf *tw;
    if (get_twiddles(nfft, &tw)) return -1;
    if (pseg_d) HIPCHK(hipMemsetAsync(pseg_d, 0, sizeof(double) * (size_t)nframes, g.stream));
#define L_(NN)                                                                                       \
    {                                                                                                \
        using C = WgCfg<NN>;                                                                         \
        int64_t G, fpg;                                                                              \
        int blocks;                                                                                  \
        run_partition<NN>(nframes, &G, &fpg, &blocks);                                               \
        if (dtype == SP_DTYPE_C64)                                                                   \
            hipLaunchKernelGGL((k_stft<NN, true>), dim3(blocks), dim3(C::WG), C::lds_bytes(1), g.stream, x, win_d, \
                               hop, nframes, fpg, mean_d, tw, sided, (float)amp, out_kind, out_d, pseg_d); \
        else                                                                                         \
            hipLaunchKernelGGL((k_stft<NN, false>), dim3(blocks), dim3(C::WG), C::lds_bytes(1), g.stream, x, win_d, \
                               hop, nframes, fpg, mean_d, tw, sided, (float)amp, out_kind, out_d, pseg_d); \
    }
    SP_DISPATCH_N(nfft, L_)
#undef L_
    HIPCHK(hipGetLastError());
    return 0;
}

template <typename E> int dev_transpose(const E *in, E *out, int64_t rows, int64_t cols) {
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    hipLaunchKernelGGL((k_transpose<E>), grid, dim3(32, 8), 0, g.stream, in, out, rows, cols);
    HIPCHK(hipGetLastError());
    return 0;
}

__global__ void k_set_means(float *f, const double *src, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) f[i] = (float)src[i];
}

__global__ void k_xcorr_norm(double *mom /*[8]: m1 d[0..3], m2 d[4..7]*/, int64_t n, double *out /*[4]*/) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double m1 = mom[0], m2 = mom[4];
        const double v1 = mom[2] / (double)n - m1 * m1, v2 = mom[6] / (double)n - m2 * m2;
        out[0] = m1;
        out[1] = m2;
        out[2] = 1.0 / ((double)n * sqrt(v1 > 0 ? v1 : 0) * sqrt(v2 > 0 ? v2 : 0));
        out[3] = 0;
    }
}

// Small host tables (windows, filter spectra) live in a content-keyed device cache: a table is uploaded
// once into its own allocation and never overwritten, so asynchronous (mem=1) callers can reuse the
// host buffer immediately and repeated calls with the same window cost no copy and no synchronisation.
struct TableEntry {
    void *dev;
    size_t bytes;
};
std::map<uint64_t, TableEntry> g_tables;

uint64_t fnv1a(const void *p, size_t n, uint64_t h) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; ++i) {
        h ^= b[i];
        h *= 1099511628211ull;
    }
    return h;
}

void tables_release() {
    for (auto &kv : g_tables) (void)hipFree(kv.second.dev);
    g_tables.clear();
}

// returns the cached device copy of `host` (bytes long); *fresh = true when it was just created
int get_table(uint64_t kind, const void *host, size_t bytes, void **dev, bool *fresh) {
    const uint64_t key = fnv1a(host, bytes, 1469598103934665603ull ^ (kind * 0x9E3779B97F4A7C15ull) ^ bytes);
    auto it = g_tables.find(key);
    if (it != g_tables.end() && it->second.bytes == bytes) {
        *dev = it->second.dev;
        if (fresh) *fresh = false;
        return 0;
    }
    if (g_tables.size() >= 64) {
        HIPCHK(hipStreamSynchronize(g.stream));
        tables_release();
    }
    void *d = nullptr;
    HIPCHK(hipMalloc(&d, bytes));
    HIPCHK(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
    g_tables[key] = TableEntry{d, bytes};
    *dev = d;
    if (fresh) *fresh = true;
    return 0;
}

__global__ void k_set2(float *dst, float a, float b) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        dst[0] = a;
        dst[1] = b;
    }
}

int next_pow2(int64_t n) {
    int64_t p = 1;
    while (p < n) p <<= 1;
    return (int)p;
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
    std::lock_guard<std::mutex> lk(g.mu);
    if (!g.ready) return;
    (void)hipDeviceSynchronize();
    for (auto &kv : g.twiddles) (void)hipFree(kv.second);
    g.twiddles.clear();
    g.in0.release();
    g.in1.release();
    g.out0.release();
    g.work.release();
    g.small.release();
    g_means.release();
    tables_release();
    g.ready = false;
}

int sp_set_stream(void *hip_stream) {
    if (ensure_init()) return -1;
    g.stream = (hipStream_t)hip_stream;
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
    std::lock_guard<std::mutex> lk(g.mu);
    const size_t esz = x_dtype == SP_DTYPE_C64 ? 8 : 4;
    const void *xd = x;
    if (!mem) {
        if (g.in0.ensure(esz * (size_t)n)) return -1;
        HIPCHK(hipMemcpyAsync(g.in0.p, x, esz * (size_t)n, hipMemcpyHostToDevice, g.stream));
        xd = g.in0.p;
    }
    MeanBuf mb;
    if (get_meanbuf(1, &mb)) return -1;
    if (dev_moments(xd, x_dtype, n, mb.d, nullptr)) return -1;
    double h[4];
    HIPCHK(hipMemcpyAsync(h, mb.d, sizeof h, hipMemcpyDeviceToHost, g.stream));
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
    std::lock_guard<std::mutex> lk(g.mu);
    if (!(is_pow2(n) && n >= 2 && n <= SP_MAX_WG_FFT)) {
        if (n == 1) {
            if (in != out) {
                if (mem) HIPCHK(hipMemcpyAsync(out, in, 8 * (size_t)batch, hipMemcpyDeviceToDevice, g.stream));
                else memcpy(out, in, 8 * (size_t)batch);
            }
            return 0;
        }
        return fail("sp_fft_c2c: n=%lld not supported yet (powers of two up to %d)", (long long)n, SP_MAX_WG_FFT);
    }
    const size_t bytes = sizeof(cf) * (size_t)n * (size_t)batch;
    const cf *din = (const cf *)in;
    cf *dout = (cf *)out;
    if (!mem) {
        if (g.in0.ensure(bytes)) return -1;
        HIPCHK(hipMemcpyAsync(g.in0.p, in, bytes, hipMemcpyHostToDevice, g.stream));
        din = (const cf *)g.in0.p;
        dout = (cf *)g.in0.p;
    }
    if (dev_fft_pow2_wg(din, dout, (int)n, batch, direction > 0)) return -1;
    if (!mem) {
        HIPCHK(hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

static int check_frames(const char *who, int64_t nsig, int nfft, int hop, int64_t nframes) {
    if (nfft < 2 || hop < 1 || nframes < 1) return fail("%s: bad nfft/hop/nframes", who);
    if ((nframes - 1) * (int64_t)hop + nfft > nsig)
        return fail("%s: %lld frames of %d with hop %d need %lld samples, signal has %lld", who, (long long)nframes, nfft,
                    hop, (long long)((nframes - 1) * (int64_t)hop + nfft), (long long)nsig);
    if (!(is_pow2(nfft) && nfft <= SP_MAX_WG_FFT))
        return fail("%s: nfft=%d not supported yet (powers of two up to %d)", who, nfft, SP_MAX_WG_FFT);
    return 0;
}

int sp_welch_psd(const void *x, int x_dtype, int64_t nsig, const float *win, int nfft, int hop, int64_t nframes,
                 int want_mean, double mean_re, double mean_im, int sided, double scale, double *pxx_out, int mem) {
    if (ensure_init()) return -1;
    if (check_frames("sp_welch_psd", nsig, nfft, hop, nframes)) return -1;
    if (sided < 1 || sided > 3) return fail("sp_welch_psd: bad sided");
    std::lock_guard<std::mutex> lk(g.mu);
    const size_t esz = x_dtype == SP_DTYPE_C64 ? 8 : 4;
    const void *xd = x;
    if (!mem) {
        if (g.in0.ensure(esz * (size_t)nsig)) return -1;
        HIPCHK(hipMemcpyAsync(g.in0.p, x, esz * (size_t)nsig, hipMemcpyHostToDevice, g.stream));
        xd = g.in0.p;
    }
    void *win_d;
    if (get_table(1, win, sizeof(float) * (size_t)nfft, &win_d, nullptr)) return -1;
    MeanBuf mb;
    if (get_meanbuf(1, &mb)) return -1;
    if (want_mean) {
        if (dev_moments(xd, x_dtype, nsig, mb.d, mb.f)) return -1;
    } else {
        hipLaunchKernelGGL(k_set2, dim3(1), dim3(64), 0, g.stream, mb.f, (float)mean_re, (float)mean_im);
    }
    const int nb = sided == SP_SIDED_ONE ? nfft / 2 : nfft;
    double *out_d = pxx_out;
    if (!mem) {
        if (g.out0.ensure(sizeof(double) * (size_t)nb)) return -1;
        out_d = (double *)g.out0.p;
    }
    if (dev_welch_psd(xd, x_dtype, (const float *)win_d, nfft, hop, nframes, mb.f, sided, scale, out_d)) return -1;
    if (!mem) {
        HIPCHK(hipMemcpyAsync(pxx_out, out_d, sizeof(double) * (size_t)nb, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

int sp_welch_csd(const void *x, const void *y, int dtype, int64_t nsig, int nch, int64_t y_ld, const float *win,
                 int nfft, int hop, int64_t nframes, int want_mean, const double *mean_x, const double *mean_y,
                 int sided, double scale, double *pxx, double *pyy, double *pxy, int mem) {
    if (ensure_init()) return -1;
    if (check_frames("sp_welch_csd", nsig, nfft, hop, nframes)) return -1;
    if (nch < 1 || y_ld < nsig) return fail("sp_welch_csd: bad nch / y_ld");
    if (sided < 1 || sided > 3) return fail("sp_welch_csd: bad sided");
    std::lock_guard<std::mutex> lk(g.mu);
    const size_t esz = dtype == SP_DTYPE_C64 ? 8 : 4;
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
    MeanBuf mb;
    if (get_meanbuf(nch, &mb)) return -1;
    if (want_mean) {
        if (dev_moments(xd, dtype, nsig, mb.d, mb.f)) return -1;
        for (int c = 0; c < nch; ++c)
            if (dev_moments((const char *)yd + esz * (size_t)y_ld * (size_t)c, dtype, nsig, mb.d + 4 * (c + 1),
                            mb.f + 2 * (c + 1)))
                return -1;
    } else {
        std::vector<float> m(2 * (size_t)(nch + 1), 0.f);
        if (mean_x) { m[0] = (float)mean_x[0]; m[1] = (float)mean_x[1]; }
        if (mean_y) for (int c = 0; c < 2 * nch; ++c) m[2 + c] = (float)mean_y[c];
        HIPCHK(hipStreamSynchronize(g.stream));   // mb.f may still be read by an earlier call
        HIPCHK(hipMemcpy(mb.f, m.data(), sizeof(float) * m.size(), hipMemcpyHostToDevice));
    }
    const size_t nb = sided == SP_SIDED_ONE ? nfft / 2 : nfft;
    double *pxx_d = pxx, *pyy_d = pyy, *pxy_d = pxy;
    if (!mem) {
        if (g.out0.ensure(sizeof(double) * nb * (1 + 3 * (size_t)nch))) return -1;
        pxx_d = (double *)g.out0.p;
        pyy_d = pxx_d + nb;
        pxy_d = pyy_d + nb * nch;
    }
    if (dev_welch_csd(xd, yd, dtype, nch, y_ld, (const float *)win_d, nfft, hop, nframes, mb.f, mb.f + 2, sided,
                      scale, pxx_d, pyy_d, pxy_d))
        return -1;
    if (!mem) {
        HIPCHK(hipMemcpyAsync(pxx, pxx_d, sizeof(double) * nb, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipMemcpyAsync(pyy, pyy_d, sizeof(double) * nb * nch, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipMemcpyAsync(pxy, pxy_d, sizeof(double) * nb * nch * 2, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

int sp_csd_matrix(const float *x, int nch, int64_t nsig, int64_t x_ld, const float *win, int nfft, int hop,
                  int64_t nframes, int want_mean, double scale, double *g_out, int mem) {
    (void)x; (void)nch; (void)nsig; (void)x_ld; (void)win; (void)nfft; (void)hop; (void)nframes; (void)want_mean;
    (void)scale; (void)g_out; (void)mem;
    return fail("sp_csd_matrix: not implemented yet");
}

int sp_stft(const void *x, int x_dtype, int64_t nsig, const float *win, int nfft, int hop, int64_t nframes,
            int want_mean, double mean_re, double mean_im, int sided, double amp_scale, int out_kind, int out_major,
            void *out, double *pseg_out, int mem) {
    if (ensure_init()) return -1;
    if (check_frames("sp_stft", nsig, nfft, hop, nframes)) return -1;
    if (sided < 1 || sided > 3) return fail("sp_stft: bad sided");
    std::lock_guard<std::mutex> lk(g.mu);
    const size_t esz = x_dtype == SP_DTYPE_C64 ? 8 : 4;
    const size_t osz = out_kind ? 4 : 8;
    const size_t nb = sided == SP_SIDED_ONE ? nfft / 2 : nfft;
    const size_t obytes = osz * nb * (size_t)nframes;
    const void *xd = x;
    if (!mem) {
        if (g.in0.ensure(esz * (size_t)nsig)) return -1;
        HIPCHK(hipMemcpyAsync(g.in0.p, x, esz * (size_t)nsig, hipMemcpyHostToDevice, g.stream));
        xd = g.in0.p;
    }
    void *win_d;
    if (get_table(1, win, sizeof(float) * (size_t)nfft, &win_d, nullptr)) return -1;
    MeanBuf mb;
    if (get_meanbuf(1, &mb)) return -1;
    if (want_mean) {
        if (dev_moments(xd, x_dtype, nsig, mb.d, mb.f)) return -1;
    } else {
        hipLaunchKernelGGL(k_set2, dim3(1), dim3(64), 0, g.stream, mb.f, (float)mean_re, (float)mean_im);
    }
    // frame-major result goes to `fm`; bin-major needs a transpose into `fin`
    void *fin = out;
    double *pseg_d = pseg_out;
    if (!mem) {
        if (g.out0.ensure(obytes + (pseg_out ? sizeof(double) * (size_t)nframes + 16 : 0))) return -1;
        fin = g.out0.p;
        if (pseg_out) pseg_d = (double *)((char *)g.out0.p + ((obytes + 15) & ~(size_t)15));
    }
    void *fm = fin;
    if (out_major == 1) {
        if (g.work.ensure(obytes)) return -1;
        fm = g.work.p;
    }
    if (dev_stft(xd, x_dtype, (const float *)win_d, nfft, hop, nframes, mb.f, sided, amp_scale, out_kind, fm, pseg_d))
        return -1;
    if (out_major == 1) {
        if (out_kind) {
            if (dev_transpose<float>((const float *)fm, (float *)fin, nframes, (int64_t)nb)) return -1;
        } else {
            if (dev_transpose<cf>((const cf *)fm, (cf *)fin, nframes, (int64_t)nb)) return -1;
        }
    }
    if (!mem) {
        HIPCHK(hipMemcpyAsync(out, fin, obytes, hipMemcpyDeviceToHost, g.stream));
        if (pseg_out)
            HIPCHK(hipMemcpyAsync(pseg_out, pseg_d, sizeof(double) * (size_t)nframes, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

int sp_hilbert(const float *x, int64_t n_in, int64_t x_ld, int64_t nfft, int64_t batch, void *out, int mem) {
    if (ensure_init()) return -1;
    if (n_in < 1 || nfft < 1 || batch < 1 || x_ld < n_in) return fail("sp_hilbert: bad sizes");
    if (!(is_pow2(nfft) && nfft >= 2 && nfft <= SP_MAX_WG_FFT))
        return fail("sp_hilbert: nfft=%lld not supported yet (powers of two up to %d)", (long long)nfft, SP_MAX_WG_FFT);
    std::lock_guard<std::mutex> lk(g.mu);
    const int64_t nuse = n_in < nfft ? n_in : nfft;
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
    const cf *tw;
    if (get_twiddles(nfft, &tw)) return -1;
#define L_(NN)                                                                                       \
    {                                                                                                \
        using C = WgCfg<NN>;                                                                         \
        const int blocks = strided_blocks<NN>(batch);                                                \
        hipLaunchKernelGGL((k_hilbert<NN>), dim3(blocks), dim3(C::WG), C::lds_bytes(1), g.stream, xd, nuse, x_ld, \
                           batch, tw, od);                                                           \
    }
    SP_DISPATCH_N((int)nfft, L_)
#undef L_
    HIPCHK(hipGetLastError());
    if (!mem) {
        HIPCHK(hipMemcpyAsync(out, od, obytes, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

int sp_xcorr(const float *x1, const float *x2, int64_t n, float *co_out, int mem) {
    if (ensure_init()) return -1;
    if (n < 1) return fail("sp_xcorr: n must be positive");
    const int L = next_pow2(2 * n);
    if (L > SP_MAX_WG_FFT) return fail("sp_xcorr: n=%lld not supported yet (n <= %d)", (long long)n, SP_MAX_WG_FFT / 2);
    std::lock_guard<std::mutex> lk(g.mu);
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
    MeanBuf mb;
    if (get_meanbuf(3, &mb)) return -1;
    if (dev_moments(a, SP_DTYPE_F32, n, mb.d, nullptr)) return -1;
    if (dev_moments(b, SP_DTYPE_F32, n, mb.d + 4, nullptr)) return -1;
    hipLaunchKernelGGL(k_xcorr_norm, dim3(1), dim3(64), 0, g.stream, mb.d, n, mb.d + 8);
    const cf *tw;
    if (get_twiddles(L, &tw)) return -1;
    const int Ln = L < 2 ? 2 : L;
#define L_(NN)                                                                                       \
    {                                                                                                \
        using C = WgCfg<NN>;                                                                         \
        hipLaunchKernelGGL((k_xcorr<NN>), dim3(1), dim3(C::WG), C::lds_bytes(1), g.stream, a, b, n, mb.d + 8, tw, od); \
    }
    SP_DISPATCH_N(Ln, L_)
#undef L_
    HIPCHK(hipGetLastError());
    if (!mem) {
        HIPCHK(hipMemcpyAsync(co_out, od, obytes, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

int sp_fftfilt(const float *h, int ntaps, const float *x, int64_t n, int nfft, float *y, int mem) {
    if (ensure_init()) return -1;
    if (ntaps < 1 || n < 1) return fail("sp_fftfilt: bad sizes");
    if (nfft == 0) {
        nfft = next_pow2(8 * (int64_t)ntaps);
        if (nfft < 1024) nfft = 1024;
        if (nfft > SP_MAX_WG_FFT) nfft = SP_MAX_WG_FFT;
    }
    if (!(is_pow2(nfft) && nfft >= 2 && nfft <= SP_MAX_WG_FFT) || nfft < 2 * (ntaps - 1) || nfft <= ntaps - 1)
        return fail("sp_fftfilt: nfft=%d must be a power of two with 2*(ntaps-1) <= nfft <= %d", nfft, SP_MAX_WG_FFT);
    std::lock_guard<std::mutex> lk(g.mu);
    // Hs = FFT(h zero-padded)/nfft, cached per (taps, nfft)
    std::vector<cf> hp((size_t)nfft, make_float2(0.f, 0.f));
    for (int i = 0; i < ntaps; ++i) hp[(size_t)i] = make_float2(h[i] / (float)nfft, 0.f);
    void *H_d;
    bool fresh = false;
    if (get_table(2, hp.data(), sizeof(cf) * (size_t)nfft, &H_d, &fresh)) return -1;
    if (fresh && dev_fft_pow2_wg((const cf *)H_d, (cf *)H_d, nfft, 1, 0)) return -1;
    const float *xd = x;
    float *yd = y;
    const size_t bytes = sizeof(float) * (size_t)n;
    if (!mem) {
        if (g.in0.ensure(bytes) || g.out0.ensure(bytes)) return -1;
        HIPCHK(hipMemcpyAsync(g.in0.p, x, bytes, hipMemcpyHostToDevice, g.stream));
        xd = (const float *)g.in0.p;
        yd = (float *)g.out0.p;
    }
    const cf *tw;
    if (get_twiddles(nfft, &tw)) return -1;
    const int64_t L = nfft - (ntaps - 1);
    const int64_t npairs = ((n + L - 1) / L + 1) / 2;
#define L_(NN)                                                                                       \
    {                                                                                                \
        using C = WgCfg<NN>;                                                                         \
        const int blocks = strided_blocks<NN>(npairs);                                               \
        hipLaunchKernelGGL((k_fftfilt<NN>), dim3(blocks), dim3(C::WG), C::lds_bytes(1), g.stream, xd, n, ntaps, \
                           (const cf *)H_d, tw, yd);                                                 \
    }
    SP_DISPATCH_N(nfft, L_)
#undef L_
    HIPCHK(hipGetLastError());
    if (!mem) {
        HIPCHK(hipMemcpyAsync(y, yd, bytes, hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
    }
    return 0;
}

}   // extern "C"
