/* spectral.h -- C ABI of libspectral.so: MI355X (gfx950) spectral-analysis kernels.
 *
 * This is the drop-in boundary underneath the Python modules in pyfft_amd/ that mirror
 * gmweir/PYFFT's numpy-array function signatures.  The reference has no FFI of its own
 * (it is pure Python over numpy.fft); each entry point below names the reference call
 * site(s) whose arithmetic it replaces.  Reference paths are relative to the reference
 * checkout (file:line).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes only.  Caller owns every buffer.
 *  - `mem`: 0 = all data pointers are host memory (library stages through its own device
 *    scratch, synchronously); 1 = all data pointers are device memory on the current
 *    device (nothing is copied; work is enqueued on the stream set by sp_set_stream and
 *    the call returns after enqueueing -- small parameter tables (window, filter taps)
 *    are ALWAYS host pointers.
 *  - complex = interleaved float re,im (numpy complex64).  Reduced spectra (Welch
 *    accumulators) are returned in double.
 *  - return value: 0 = ok, <0 = error; sp_last_error() gives the message (thread-local).
 *  - one global context per process / one device per process (multi-GPU = one process per
 *    GPU, torch.distributed/RCCL reduces the accumulators; see pyfft_amd/dist.py).
 *  - Transform convention (dft.py:108-133, :242-290): forward unnormalised e^{-j2pi nk/N},
 *    inverse scaled 1/N.
 */
#ifndef SPECTRAL_H
#define SPECTRAL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SP_DTYPE_F32 0 /* real float32 samples    */
#define SP_DTYPE_C64 1 /* complex64 samples (re,im) */

#define SP_SIDED_ONE 1 /* reference one-sided: bins [0,N/2), x2 on [1:-1] (Nyquist dropped; Q1) */
#define SP_SIDED_TWO 2 /* two-sided, fftshift-ed                                               */
#define SP_SIDED_RAW 3 /* two-sided, natural FFT order, no doubling                            */
#define SP_SIDED_HALF 4 /* bins 0..nfft/2 (numpy rfft layout), no doubling                        */

/* ---- context ------------------------------------------------------------------------ */
int sp_init(int device_id);          /* select device, create context (idempotent)          */
void sp_shutdown(void);              /* free plan cache + scratch                           */
const char *sp_last_error(void);
/* Concurrency contract: one global context under one lock -- calls from several threads serialise.  All launches go to
 * ONE stream (the one set last); scratch buffers, the device-table cache and the pending state between sp_welch_accum and
 * sp_welch_finish are shared by all calls.  sp_set_stream orders the new stream behind the work already queued on the
 * old one (event), so consecutive calls from different streams are safe; truly concurrent use from two streams is not
 * supported.  The table cache (windows, FFT(window), filter spectra; 64 entries, LRU) never evicts a table the current
 * call obtained nor the tables a pending sp_welch_accum holds. */
int sp_set_stream(void *hip_stream); /* hipStream_t for mem=1 calls (NULL = default stream) */
int sp_synchronize(void);            /* wait for the library's stream                       */
int sp_version(void);
int sp_max_wg_fft(void);             /* largest power-of-two FFT done inside one workgroup  */
/* Measurement hook: when enabled, the dominant kernel of each Welch call (k_welch) is bracketed by HIP events
 * recorded on the launch stream; sp_profile_last_ms() waits for them and returns the kernel's duration. */
int sp_profile_enable(int on);
int sp_profile_last_ms(double *ms);
const char *sp_profile_last_kernel(void); /* name of the kernel the last Welch call dispatched */
/* device properties used by the host to size grids: out[0]=CU count, out[1]=LDS bytes/CU,
 * out[2]=clock kHz, out[3]=wavefront size */
int sp_device_info(int64_t out[4]);

/* ---- A7: fftanal.fft / .ifft  (fft_analysis.py:2096-2116 -> np.fft.fft/ifft) ---------- */
/* batch x n-point C2C transforms, rows contiguous.  n: any length >= 1 (powers of two up
 * to sp_max_wg_fft() run in one workgroup; longer powers of two use the multi-pass
 * four-step path; other lengths use Bluestein's chirp-z on top of those).
 * direction: -1 forward, +1 inverse (scaled 1/n). in == out allowed. */
int sp_fft_c2c(const void *in, void *out, int64_t n, int64_t batch, int direction, int mem);

/* ---- A3+A4: fftanal.fft_win -> Pstft -> averagewins (fft_analysis.py:2126-2203,
 *      :1944-1990), the fused Welch PSD: per frame g, X_g = FFT(win * (x[g*hop : g*hop+nfft] - trend));
 *      pxx[k] = scale/nframes * sum_g |X_g[k]|^2 with the sidedness permutation/doubling.
 *      nfft: any length >= 2.  Powers of two up to sp_max_wg_fft() and other lengths up to half of it run in ONE fused
 *      kernel (Bluestein inside the workgroup); longer segments -- the reference's default regime, Navr = 8 ->
 *      nwins = floor(nsig / 4.5), fft_analysis.py:2412-2418 -- go through the multi-kernel long-segment path (pack ->
 *      batched multi-pass FFT / chirp-z -> float64 accumulate; k_long.hip), up to a 2^26-point transform.  The same
 *      holds for sp_welch_csd, sp_stft and sp_stft_cog.
 *      detrend (global detrend over x[0:nsig], fft_analysis.py:2148, :2539-2549):
 *        SP_DETREND_CONST  (0) subtract the given constant mean_re + i mean_im (0,0 = no detrend),
 *        SP_DETREND_MEAN   (1) the library computes and subtracts the mean (one extra pass over x),
 *        SP_DETREND_LINEAR (2) the library fits and subtracts the least-squares line,
 *        SP_DETREND_SEGMEAN (3) every segment's own mean is removed before the window (the per-segment detrend of the
 *                              matplotlib.mlab estimators behind fft_analysis.psd/csd/coh, :1060-1155); sp_welch_psd,
 *                              sp_welch_csd and sp_stft (fft_win's detrendwin=True),
 *        SP_DETREND_SEGLINEAR (4) the same with every segment's own least-squares line.
 *      nbins = Nnyquist for SP_SIDED_ONE (nfft/2, or (nfft+1)/2 when odd), nfft otherwise. */
#define SP_DETREND_CONST 0
#define SP_DETREND_MEAN 1
#define SP_DETREND_LINEAR 2
#define SP_DETREND_SEGMEAN 3
#define SP_DETREND_SEGLINEAR 4
int sp_welch_psd(const void *x, int x_dtype, int64_t nsig, const float *win, int nfft, int hop,
                 int64_t nframes, int detrend, double mean_re, double mean_im, int sided,
                 double scale, double *pxx_out, int mem);

/* ---- the same path split in two for segment-sharded multi-process runs (one process per GPU): every process
 *      accumulates its own frames against a local estimate of the mean, the processes all-reduce sum_out (2 doubles)
 *      to get the global mean of the stream, and each finishes with it; the finished spectra (scaled by
 *      scale/frames_total) then add up to the Welch PSD of the whole stream.  Needs a power-of-two nfft in
 *      [256, sp_max_wg_fft()] with hop = nfft/4, nfft/2 or nfft.  One accumulation may be pending at a time; x must
 *      stay valid until sp_welch_finish when mem=1.
 *      nmean: this shard's own samples x[0:nmean] (halo excluded) -> sum_out[2] = sum of them.
 *      mean: [2] doubles (host if mem=0, device if mem=1), or NULL = the shard's own mean sum_out/nmean. */
int sp_welch_accum(const void *x, int x_dtype, int64_t nsig, const float *win, int nfft, int hop, int64_t nframes,
                   int64_t nmean, double *sum_out, int mem);
int sp_welch_finish(const double *mean, int64_t frames_total, int sided, double scale, double *pxx_out, int mem);
/*      One-collective form of the same split.  sp_welch_export leaves this shard's ADDITIVE state in
 *      state[5*nfft + 8] (doubles): sum|X|^2, sum X and conj(mu0) sum X per bin (spectra taken against the shard's
 *      own mean estimate mu0), M mu0, M |mu0|^2, the sum of its nmean own samples, M and nmean.  The states of all
 *      shards are summed with ONE all-reduce and sp_welch_apply turns the sum into the PSD of the whole stream
 *      detrended by its global mean (same output conventions as sp_welch_finish). */
int sp_welch_export(const void *x, int x_dtype, int64_t nsig, const float *win, int nfft, int hop,
                    int64_t nframes, int64_t nmean, double *state, int mem);
int sp_welch_apply(const double *state, const float *win, int nfft, int64_t frames_total, int sided, double scale,
                   double *pxx_out, int mem);

/* ---- the same path as a STREAM of steps, on one GPU or across the GPUs of one node, without the host in the loop (device
 *      pointers only).  SURVEY section 8b sketched `sp_init(device_count, device_ids)` + an `ngpu` argument; the MI355X-native
 *      form is one process per GPU with the library owning an RCCL communicator.
 *      Communicator (optional): every process sp_init(its device); rank 0 calls sp_comm_unique_id and hands the
 *      SP_COMM_ID_BYTES to the others by any means (the Python layer: one torch.distributed broadcast; a C host: MPI / a file /
 *      a socket); all call sp_comm_init(id, world, rank) (collective).  RCCL is resolved at run time (dlopen "librccl.so.1":
 *      the copy already in the process if there is one), so the library loads without it.
 *      sp_welch_dist_submit(step k): this shard's main kernel on the launch stream; its epilogue on the library's own stream
 *      behind an event -- without communicator the finished PSD, with one the shard's additive state (sp_welch_export), ONE
 *      ncclAllReduce (sum, double, 5 nfft + 8 values) and the sp_welch_apply, folded into the next step's epilogue launch --
 *      so the epilogue and the collective run BESIDE the next step's main kernel.  pxx_out[nbins] (device) receives THIS
 *      step's PSD of the whole stream; it is valid on the launch stream once a later call has reported it: *ndone = how many
 *      earlier submits' outputs became valid with this call, in submit order (without communicator: step k-1 at submit k;
 *      with: step k-2).  sp_welch_dist_flush reports the rest.  x, win contents and pxx_out of a step must stay alive until
 *      it is reported.  K submits + one flush = K Welch PSDs; no host synchronisation.  frames_total = frames of the WHOLE
 *      stream (the normalisation); nmean = this shard's own samples (halo excluded).  sp_welch_psd_dist = submit + flush.
 *      Shapes: as sp_welch_export.  Reference path: the same as sp_welch_psd (fft_analysis.py:2126-2203, :1944-1990) over
 *      segments dealt out to the ranks.
 *      With a communicator of more than one rank the library asks RCCL for at most 4 workgroups per collective
 *      (ncclCommInitRankConfig; SP_DIST_RCCL_CTAS) and partitions its main kernels over all but 8 CUs (SP_DIST_RESERVE_CUS), so
 *      that RCCL's kernel -- which cannot share a CU with the main kernel -- runs beside it instead of delaying the next one. */
#define SP_COMM_ID_BYTES 128
int sp_comm_unique_id(void *id_out /* SP_COMM_ID_BYTES */);
int sp_comm_init(const void *id, int world, int rank);
int sp_comm_info(int out[2] /* world (0 = no communicator), rank */);
int sp_comm_destroy(void);
int sp_welch_dist_submit(const void *x, int x_dtype, int64_t nsig, const float *win, int nfft, int hop, int64_t nframes,
                         int64_t nmean, int64_t frames_total, int sided, double scale, double *pxx_out, int *ndone);
int sp_welch_dist_flush(int *ndone);
int sp_welch_psd_dist(const void *x, int x_dtype, int64_t nsig, const float *win, int nfft, int hop, int64_t nframes,
                      int64_t nmean, int64_t frames_total, int sided, double scale, double *pxx_out);

/* ---- A5: fft_pwelch numeric core (fft_analysis.py:339-446): reference x against nch
 *      channels y[c][0:nsig] (channel-major, row stride y_ld samples).
 *      pxx[nbins], pyy[nch][nbins], pxy[nch][nbins] complex (re,im doubles) = Y_c * conj(X)
 *      (function-path conjugation, :393; the class path's X*conj(Y), :1960, is its conjugate).
 *      detrend as above (per signal); SP_DETREND_CONST subtracts mean_x / mean_y[nch] (NULL = 0). */
int sp_welch_csd(const void *x, const void *y, int dtype, int64_t nsig, int nch, int64_t y_ld,
                 const float *win, int nfft, int hop, int64_t nframes, int detrend,
                 const double *mean_x /*[2]*/, const double *mean_y /*[nch][2]*/, int sided,
                 double scale, double *pxx, double *pyy, double *pxy, int mem);

/* ---- cfg5: full cross-spectral-density matrix of nch real channels x[c][0:nsig]
 *      (generalises the ref x channel loop fft_analysis.py:387-393 / HeatPulse_Funcs.py:576-583).
 *      g_out[nfft/2+1][nch][nch] complex double, = scale/nframes * sum_g X_i conj(X_j), no doubling. */
int sp_csd_matrix(const float *x, int nch, int64_t nsig, int64_t x_ld, const float *win, int nfft,
                  int hop, int64_t nframes, int detrend, double scale, double *g_out, int mem);
/*      The same with the per-channel constants to remove given by the caller (HOST array means[nch], like
 *      mean_y of sp_welch_csd): a frame-sharded CSD matrix detrends every shard with the mean of the WHOLE
 *      record (fft_analysis.py:2148 semantics), obtained from sp_channel_means + an all-reduce. */
int sp_csd_matrix_means(const float *x, int nch, int64_t nsig, int64_t x_ld, const float *win, int nfft,
                        int hop, int64_t nframes, const double *means, double scale, double *g_out, int mem);
/*      means_out[c] = mean of x[c][0:nsig]  (nch real channels, row stride x_ld); means_out follows `mem`. */
int sp_channel_means(const float *x, int nch, int64_t nsig, int64_t x_ld, double *means_out, int mem);

/* ---- A8/A9: spectrogram.stft -> fftanal.fft_win (spectrogram.py:140-168, fft_analysis.py:2126-2203)
 *      and spectrogram.specgram (spectrogram.py:91-112).
 *      out_kind 0: complex64 amp_scale * X_g[k] (sqrt(2) on [1:-1] for SP_SIDED_ONE);
 *      out_kind 1: float32 power amp_scale * |X_g[k]|^2 (no doubling).
 *      out_major 0: [nframes][nbins] (fftanal Xseg); 1: [nbins][nframes] (specgram).
 *      pseg_out (may be NULL): float64[nframes] = trapz(|win*(x-mean)|^2) with unit spacing (:2174; host
 *      multiplies by dt and divides by S2). */
int sp_stft(const void *x, int x_dtype, int64_t nsig, const float *win, int nfft, int hop, int64_t nframes,
            int detrend, double mean_re, double mean_im, int sided, double amp_scale, int out_kind,
            int out_major, void *out, double *pseg_out, int mem);

/* ---- N3: Doppler.cog applied per STFT frame (Doppler.py:43-58; the loop body of cogspec, Doppler.py:73-81):
 *      cog_out[g] = sum_k f_k |X_g[k]|^2 / sum_k |X_g[k]|^2 over the two-sided spectrum of frame g, f_k = fftfreq(nfft, 1/fs),
 *      restricted to fmin <= |f_k| <= fmax (fmin = 0, fmax >= fs/2: every bin); 0 where the band holds no power.  The
 *      spectrogram is never written: the moments are reduced inside the transform kernel.  float64[nframes], follows `mem`.
 *      Frame g = win * detrended x[g*hop : g*hop+nfft]; a boxcar window with SP_DETREND_CONST (0,0) is cog(x[frame], fs). */
int sp_stft_cog(const void *x, int x_dtype, int64_t nsig, const float *win, int nfft, int hop, int64_t nframes,
                int detrend, double mean_re, double mean_im, double fs, double fmin, double fmax, double *cog_out, int mem);

/* ---- A10: hilbert.hilbert / hilbert_1d (hilbert.py:22-112): rows of n_in real samples (row stride
 *      x_ld), transform length nfft (zero-pad / truncate like np.fft.fft(n=nfft)), one-sided mask with the
 *      reference's odd-length convention (bin nyq untouched), inverse; out[batch][nfft] complex64. */
int sp_hilbert(const float *x, int64_t n_in, int64_t x_ld, int64_t nfft, int64_t batch, void *out, int mem);

/* ---- A5, nT-model branch of fft_pwelch (fft_analysis.py:169-176, :346-393: a one-window model signal against every
 *      window of the long channels): out[ch][n] = sum_g detrended(x[ch][g*hop + n]), n < nfft -- the time-domain sum of
 *      all frames, from which sum_g FFT(win * frame_g) = FFT(win * out) follows by linearity (no spectrum is written).
 *      nch channels with row stride x_ld; detrend 0 none, 1 each channel's mean over [0:nsig], 2 its least-squares line;
 *      out float64 [nch][nfft][2] (re, im), follows `mem`. */
int sp_frame_sum(const void *x, int x_dtype, int64_t nsig, int nch, int64_t x_ld, int nfft, int hop, int64_t nframes,
                 int detrend, double *out, int mem);

/* ---- N4: frequency-domain response applied to real rows: out = IFFT(H * FFT(x, nfft)) -- the transform pair of
 *      fft_deriv (fft_analysis.py:1526-1546: `real(ifft(wavenumber * fft(sig)))`), the Hilbert kernel with the mask
 *      replaced by a table.  H: complex64[nfft], always a HOST array (like the window tables); x rows of n_in real
 *      samples (row stride x_ld), zero-padded / truncated to nfft; out[batch][nfft] complex64 (x, out follow `mem`). */
int sp_spectral_filter(const float *x, int64_t n_in, int64_t x_ld, int64_t nfft, int64_t batch, const void *H, void *out,
                       int mem);

/* ---- A11: ccf.ccf (ccf.py:66-77): normalised cross-covariance of two real length-n signals at all
 *      2n-1 lags, via zero-padded FFTs; co_out[2n-1] float32 in np.correlate(...,'full') order. */
int sp_xcorr(const float *x1, const float *x2, int64_t n, float *co_out, int mem);

/* ---- F1 (build-defined; nearest reference code filters.py:282, ccf.py:283): causal FIR
 *      y = lfilter(h, 1, x)[0:n] by overlap-save with nfft-point blocks (nfft power of two > ntaps;
 *      0 = choose). */
int sp_fftfilt(const float *h, int ntaps, const float *x, int64_t n, int nfft, float *y, int mem);

/* ---- A6 / N1: the fft_pwelch epilogue on device-resident averaged spectra (fft_analysis.py:489-648, Cxy_Cxy2 :1662-1688):
 *      complex coherence, mean-squared coherence, cross-phase, linear amplitude spectra (:526-540), and the correlations
 *      Rxx, Ryy, Rxy, iCxy = sqrt(nfft) ifft(spectrum) (one-sided input: [1:-1] halved, irfft semantics; two-sided:
 *      ifftshift first), fftshifted (:544-597), corrcoef = Rxy / sqrt(Ex Ey).  Inputs as sp_welch_csd leaves them:
 *      pxx[nb], pyy[nch][nb], pxy[nch][nb][2] float64.  `out` holds sp_csd_epilogue_doubles(nch, nb, nfft) float64:
 *        cxy[nch][nb][2] | cxy2[nch][nb] | phi[nch][nb] | lxx[nb] | lyy[nch][nb] | lxy[nch][nb] |
 *        rxx[nfft][2] | ryy[nch][nfft][2] | rxy[nch][nfft][2] | icxy[nch][nfft][2] | corrcoef[nch][nfft][2] | e[1+nch][2]
 *      (e = the zero-lag values Ex, Ey_c; imaginary parts are zero for one-sided input). */
int64_t sp_csd_epilogue_doubles(int nch, int nb, int nfft);
int sp_csd_epilogue(const double *pxx, const double *pyy, const double *pxy, int nch, int nb, int nfft, int onesided, double enbw,
                    double *out, int mem);

/* ---- F2 (build-defined): application of the second-order sections the reference only designs (notch_filter.py:19-241
 *      iirnotch / iirpeak return (b, a) and nothing in the reference applies them; its scipy application sites for
 *      other filters are filters.py:328 lfilter and :347 filtfilt).  y = scipy.signal.lfilter(b, a, x) for one biquad,
 *      b[3], a[3] float64 HOST arrays (a[0] != 0), x and y float32 [n] (follow `mem`).  The recurrence is evaluated
 *      exactly (float64 state, blocked scan of the affine state maps), not as a truncated FIR. */
int sp_biquad(const double *b, const double *a, const float *x, int64_t n, float *y, int mem);

/* ---- helper: mean of a float32 / complex64 vector in double (fft_analysis.py:2148 detrend) */
int sp_mean(const void *x, int x_dtype, int64_t n, double out[2], int mem);

#ifdef __cplusplus
}
#endif
#endif
