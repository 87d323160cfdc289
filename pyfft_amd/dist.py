"""Segment-sharded Welch PSD / CSD matrix and channel-sharded reference-vs-channels CSD across the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" =
RCCL over xGMI).  The path shards by frames -- every rank owns a contiguous range of segments of one long stream
plus a (nfft-hop)-sample halo -- and has exactly one exchange: an all-reduce of the shards' additive state (the |X|^2
accumulator plus what is needed to apply the mean of the WHOLE stream afterwards, fft_analysis.py:2148; 160 KiB at
nfft=4096: latency-bound on any topology).

    plan = shard_plan(total_samples, nfft, hop, world, rank)
    x_local = stream[plan.first_sample : plan.first_sample + plan.nsamples]          # device resident
    pxx = welch_psd_sharded(x_local, win, plan, scale)                               # same on every rank
"""
from collections import namedtuple

import numpy as np

ShardPlan = namedtuple("ShardPlan", "rank world nfft hop frames_total first_frame frames first_sample nsamples "
                                    "own_samples total_samples")


def shard_plan(total_samples, nfft, hop, world, rank):
    """Contiguous frame ranges, as even as possible.  Rank r owns the samples from its first frame's first sample up
    to the next rank's first frame (the last rank: to the end of the stream); it additionally READS the halo that
    its last frames extend into."""
    total_samples, nfft, hop = int(total_samples), int(nfft), int(hop)
    if total_samples < nfft:
        raise ValueError("stream shorter than one segment")
    M = (total_samples - nfft) // hop + 1
    if world > M:
        raise ValueError("more ranks (%d) than frames (%d)" % (world, M))
    base, extra = divmod(M, world)
    f0 = rank * base + min(rank, extra)
    nf = base + (1 if rank < extra else 0)
    first = f0 * hop
    nsamp = (nf - 1) * hop + nfft
    own_end = total_samples if rank == world - 1 else (f0 + nf) * hop
    own_start = 0 if rank == 0 else first
    return ShardPlan(rank, world, nfft, hop, M, f0, nf, first, nsamp if rank < world - 1 else total_samples - first,
                     own_end - own_start, total_samples)


def _device_backend():
    from . import engine as E

    def export(x, win, hop, frames, nmean):
        return E.welch_export(x, win, hop, frames, nmean=nmean)

    def apply(state, win, frames_total, sided, scale):
        return E.welch_apply(state, win, frames_total, sided=sided, scale=scale)
    return export, apply


def welch_psd_sharded(x_local, win, plan, scale=1.0, sided=2, group=None, backend=None, force_collective=False):
    """Welch PSD of the whole stream from this rank's shard (global-mean detrend) with ONE collective: every rank
    exports its additive state (sum|X|^2, sum X, conj(mu0) sum X per bin against its own mean estimate mu0, plus a few
    scalars: 5 nfft + 8 doubles, 160 KiB at nfft = 4096 -- latency-bound on any topology), the states are summed with
    one all_reduce, and every rank applies the global mean to the sum.  `backend` = (export, apply) callables; default:
    the HIP kernels (sp_welch_export / sp_welch_apply).  With world == 1 no collective is issued unless force_collective
    (a process group of one rank: the RCCL call path on a one-GPU box)."""
    import torch
    import torch.distributed as dist
    export, apply = backend if backend is not None else _device_backend()
    s = export(x_local, win, plan.hop, plan.frames, plan.own_samples)
    is_t = isinstance(s, torch.Tensor)
    st = s if is_t else torch.from_numpy(np.ascontiguousarray(s, dtype=np.float64))
    if plan.world > 1 or force_collective:
        dist.all_reduce(st, group=group)
    return apply(st if is_t else st.numpy(), win, plan.frames_total, sided, scale)


class WelchPipeline(object):
    """welch_psd_sharded with the collective of one step hidden behind the kernels of the next one.

    The exchange is 5 nfft + 8 doubles (160 KiB at nfft = 4096): pure latency, ~5 % of a 2^28-sample step when it sits
    between the accumulate kernel and the finish kernel.  `submit(x_local)` runs the export kernels of THIS step, starts
    its all-reduce asynchronously (torch.distributed async_op: on RCCL the collective runs on the process group's own
    stream, ordered after the export kernels) and then finishes the PREVIOUS step, whose all-reduce has had a whole
    step to complete; it returns the previous step's PSD (None on the first call).  `flush()` finishes the last one.
    K submits + one flush do exactly the work of K welch_psd_sharded calls."""

    def __init__(self, win, plan, scale=1.0, sided=2, group=None, backend=None, force_collective=False):
        self.win, self.plan, self.scale, self.sided, self.group = win, plan, scale, sided, group
        self.collective = plan.world > 1 or force_collective       # (forced: a process group of ONE rank still issues it)
        self.export, self.apply = backend if backend is not None else _device_backend()
        self._pending = None

    def _finish(self, pending):
        import torch
        st, work, is_t = pending
        if work is not None:
            work.wait()                  # RCCL: a stream-level wait; gloo: blocks the host until the sum has arrived
        return self.apply(st if is_t else st.numpy(), self.win, self.plan.frames_total, self.sided, self.scale)

    def submit(self, x_local):
        import torch
        import torch.distributed as dist
        s = self.export(x_local, self.win, self.plan.hop, self.plan.frames, self.plan.own_samples)
        is_t = isinstance(s, torch.Tensor)
        st = s if is_t else torch.from_numpy(np.ascontiguousarray(s, dtype=np.float64))
        work = dist.all_reduce(st, group=self.group, async_op=True) if self.collective else None
        prev, self._pending = self._pending, (st, work, is_t)
        return None if prev is None else self._finish(prev)

    def flush(self):
        prev, self._pending = self._pending, None
        return None if prev is None else self._finish(prev)


def native_comm_init(group=None, device=None):
    """Create the library's own RCCL communicator over the ranks of the torch.distributed group (one process per GPU):
    rank 0 draws the unique id, one broadcast_object_list hands it out, every rank joins.  After this the sharded PSD runs
    without the host in the loop (NativeWelchPipeline).  Returns (world, rank)."""
    import torch.distributed as dist
    from . import engine as E
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [E.comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    E.comm_init(box[0], world, rank, device)
    return world, rank


_ENGINE_FIFO = []        # (owner, out, x) of the library's ONE streaming engine, in submit order: kept alive until reported


class NativeWelchPipeline(object):
    """WelchPipeline with the whole step inside libspectral (sp_welch_dist_submit / sp_welch_dist_flush): one ctypes call per
    step, no torch.distributed work object, no host synchronisation.  The main kernel runs on torch's current stream; the
    epilogue (column sums + finish: ~16 us, a third of a 2^25-sample shard's step) and -- with a communicator
    (native_comm_init) -- the RCCL all-reduce of the shard state run on the library's own stream BESIDE the next step's main
    kernel.  Works without a communicator too (one GPU: the epilogue overlap alone).  Same interface as WelchPipeline, except
    that a result may arrive two submits late: submit() returns the newest of THIS pipeline's PSDs that became valid (or None),
    flush() the last one, flush_all() every outstanding one in order.  The library has one engine: several pipeline objects
    share it (their steps are reported to their owners in submit order)."""

    def __init__(self, win, plan, scale=1.0, sided=2):
        self.win, self.plan, self.scale, self.sided = win, plan, scale, sided
        self._ready = []

    @staticmethod
    def _report(n):
        for _ in range(min(n, len(_ENGINE_FIFO))):
            owner, out, _x = _ENGINE_FIFO.pop(0)
            owner._ready.append(out)

    def submit(self, x_local):
        from . import engine as E
        p = self.plan
        out, xs, nd = E.welch_dist_submit(x_local, self.win, p.hop, p.frames, p.own_samples, p.frames_total, self.sided, self.scale)
        _ENGINE_FIFO.append((self, out, xs))
        self._report(nd)
        if not self._ready:
            return None
        newest, self._ready = self._ready[-1], []
        return newest

    def flush_all(self):
        from . import engine as E
        if _ENGINE_FIFO:
            E.welch_dist_flush()
            self._report(len(_ENGINE_FIFO))
        done, self._ready = self._ready, []
        return done

    def flush(self):
        done = self.flush_all()
        return done[-1] if done else None


def welch_psd_sharded_two_step(x_local, win, plan, scale=1.0, sided=2, group=None):
    """The same result with the older split (sp_welch_accum / sp_welch_finish): all_reduce(2 doubles) of the sample sums,
    finish with the global mean, all_reduce(nbins doubles) of the finished shard spectra.  Two collectives; kept for the
    C ABI's sake and A/B tests."""
    import torch
    import torch.distributed as dist
    from . import engine as E
    st = E.welch_accum(x_local, win, plan.hop, plan.frames, nmean=plan.own_samples)
    is_t = isinstance(st, torch.Tensor)
    stt = st if is_t else torch.from_numpy(np.asarray(st, dtype=np.float64))
    if plan.world > 1:
        dist.all_reduce(stt, group=group)
    mean = stt / float(plan.total_samples)
    p = E.welch_finish(plan.nfft, mean if is_t else mean.numpy(), plan.frames_total, sided=sided, scale=scale, like=x_local)
    pt = p if isinstance(p, torch.Tensor) else torch.from_numpy(np.asarray(p, dtype=np.float64))
    if plan.world > 1:
        dist.all_reduce(pt, group=group)
    return pt if isinstance(p, torch.Tensor) else pt.numpy()


def _hermitian_pack(g, dtype):
    """upper triangle (i <= j) of G[k, i, j] as [nb, nch (nch + 1) / 2] in `dtype` (torch tensor in, torch tensor out)"""
    import torch
    nch = g.shape[-1]
    iu = torch.triu_indices(nch, nch, device=g.device)
    return g[:, iu[0], iu[1]].to(dtype).contiguous(), iu


def _hermitian_unpack(tri, iu, nch, dtype):
    import torch
    nb = tri.shape[0]
    g = torch.zeros((nb, nch, nch), dtype=dtype, device=tri.device)
    t = tri.to(dtype)
    g[:, iu[1], iu[0]] = torch.conj(t)
    g[:, iu[0], iu[1]] = t                       # the diagonal from the un-conjugated copy
    return g


def csd_matrix_sharded(x_local, win, plan, scale=1.0, group=None, backend=None, compact=None, force_collective=False):
    """Full CSD matrix (BASELINE cfg5) of a long multi-channel record from this rank's frame shard x_local[nch,
    plan.nsamples] (same ShardPlan as the PSD: contiguous frame ranges + halo).  Every channel is detrended with the
    mean of the WHOLE record: all_reduce(nch doubles) of the shard sample sums first, then each rank contracts its
    frames and the accumulator -- the only large exchange of the scope -- is summed with one all_reduce.
    compact=True sends only the Hermitian upper triangle in complex64: (nfft/2+1) nch (nch+1)/2 x 8 B, 34 MB at
    cfg5 instead of the 134 MB of the full complex128 matrix (xGMI ring all-reduce is per-link bound, so the bytes are
    the time); the HIP kernels' matrix is float32-accurate anyway and at most `world` such terms are added, so the
    returned complex128 tensor carries float32 content (~6e-8 relative per term) whatever the world size.
    compact=False reduces the full complex128 matrix.  compact=None (default): True for the HIP kernels, False when a
    `backend` is supplied (a float64 backend must not lose precision silently).  `backend` = (means, matrix) callables;
    default: the HIP kernels.  force_collective: issue the collectives in a process group of one rank too."""
    import torch
    import torch.distributed as dist
    if compact is None:
        compact = backend is None
    coll = plan.world > 1 or force_collective
    if backend is None:
        from . import engine as E
        means_fn = lambda x, n: E.channel_means(x, n)                                           # noqa: E731
        matrix_fn = lambda x, w, hop, frames, means, sc: E.csd_matrix(x, w, hop, frames, scale=sc, means=means)   # noqa: E731
    else:
        means_fn, matrix_fn = backend
    m = means_fn(x_local, plan.own_samples)                       # mean over the samples this rank owns
    is_t = isinstance(m, torch.Tensor)
    st = (m if is_t else torch.from_numpy(np.asarray(m, dtype=np.float64))) * float(plan.own_samples)
    if coll:
        dist.all_reduce(st, group=group)
    gmean = st / float(plan.total_samples)
    # local contraction normalised by the frames of the whole record, so that the shard results simply add
    g = matrix_fn(x_local, win, plan.hop, plan.frames, gmean if is_t else gmean.numpy(),
                  scale * float(plan.frames) / float(plan.frames_total))
    gt = g if isinstance(g, torch.Tensor) else torch.from_numpy(np.asarray(g, dtype=np.complex128))
    if coll:
        if compact and gt.is_complex() and gt.dim() == 3:
            tri, iu = _hermitian_pack(gt, torch.complex64)
            dist.all_reduce(torch.view_as_real(tri), group=group)
            gt = _hermitian_unpack(tri, iu, gt.shape[-1], gt.dtype)
        elif gt.is_complex():
            dist.all_reduce(torch.view_as_real(gt), group=group)
        else:
            dist.all_reduce(gt, group=group)
    return gt if isinstance(g, torch.Tensor) else gt.numpy()


def welch_csd_channel_sharded(x, y_local, win, hop, nframes, scale=1.0, sided=2, group=None, backend=None):
    """Reference signal x against many channels (fft_analysis.py:387-393): the channels are dealt out to the ranks,
    x is replicated, every signal is whole on its rank -- so there is nothing to reduce; the per-channel spectra are
    gathered.  Returns (pxx, pyy_all, pxy_all) with the channels of rank 0 first (equal channel counts per rank)."""
    import torch
    import torch.distributed as dist
    if backend is None:
        from . import engine as E
        backend = lambda a, b: E.welch_csd(a, b, win, hop, nframes, detrend=True, sided=sided, scale=scale)   # noqa: E731
    pxx, pyy, pxy = backend(x, y_local)
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if world == 1:
        return pxx, pyy, pxy
    as_t = lambda a: a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))   # noqa: E731
    outs = []
    for a in (pyy, pxy):
        t = as_t(a)
        parts = [torch.empty_like(t) for _ in range(world)]
        if t.is_complex():
            dist.all_gather([torch.view_as_real(p) for p in parts], torch.view_as_real(t), group=group)
        else:
            dist.all_gather(parts, t, group=group)
        cat = torch.cat(parts, dim=0)
        outs.append(cat if isinstance(a, torch.Tensor) else cat.numpy())
    return pxx, outs[0], outs[1]


def cog_frames_sharded(x_local, win, plan, fs, fmin=0.0, fmax=None, mean_value=None, gather=True, group=None, backend=None):
    """Centre of gravity of every frame of one long stream (Doppler.py:43-81), frames dealt out as in `shard_plan`: frames
    are independent, so the data path has NO collective -- every rank runs `engine.stft_cog` on its frames (plus halo).
    gather=True additionally all-gathers the per-frame results (8 bytes per frame) so that every rank returns the whole
    vector [frames_total]; gather=False returns the rank's own [plan.frames].  A global-mean detrend needs the mean of the
    whole stream: pass it as `mean_value` (e.g. from `welch_psd_sharded_two_step`'s sum all-reduce); default none.
    `backend(x, win, hop, frames, fs, fmin, fmax, mean_value)` -> float64 [frames]; default: the HIP kernels."""
    import torch
    import torch.distributed as dist
    if backend is None:
        from . import engine as E

        def backend(x, w, hop, frames, fs_, lo, hi, mv):
            return E.stft_cog(x, w, hop, frames, fs_, fmin=lo, fmax=hi, detrend=mv is not None, mean_value=mv)
    mine = backend(x_local, win, plan.hop, plan.frames, fs, fmin, fmax, mean_value)
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if world == 1 or not gather:
        return mine
    t = mine if isinstance(mine, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(mine, dtype=np.float64))
    # frame counts differ by at most one between ranks (shard_plan): pad to the largest, gather, trim
    base, extra = divmod(plan.frames_total, plan.world)
    width = base + (1 if extra else 0)
    padded = torch.zeros(width, dtype=torch.float64, device=t.device)
    padded[:plan.frames] = t
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    out = torch.cat([parts[r][:base + (1 if r < extra else 0)] for r in range(world)])
    return out if isinstance(mine, torch.Tensor) else out.numpy()


# ---- paths without a reduction: very long streams dealt out in pieces (north star: "very-long-stream overlap-save shard
# ... segments across the 8 GPUs"; SURVEY 8e: "replicas + concatenation", no collective on the data path) ---------------
SamplePlan = namedtuple("SamplePlan", "rank world ntaps total first last read_first nread")


def sample_shard_plan(total_samples, ntaps, world, rank):
    """Overlap-save FIR across ranks: rank r produces the outputs [first, last) of y = lfilter(h, 1, x) and READS the
    inputs [read_first, read_first + nread) = its own samples plus the ntaps-1 samples before them (the filter's memory;
    rank 0 has none: the stream starts at rest)."""
    total_samples, ntaps = int(total_samples), int(ntaps)
    if world > total_samples:
        raise ValueError("more ranks than samples")
    base, extra = divmod(total_samples, world)
    first = rank * base + min(rank, extra)
    last = first + base + (1 if rank < extra else 0)
    rf = max(0, first - (ntaps - 1))
    return SamplePlan(rank, world, ntaps, total_samples, first, last, rf, last - rf)


def fftfilt_sharded(h, x_local, plan, gather=False, group=None, backend=None):
    """This rank's piece y[plan.first:plan.last] of the causal FIR y = lfilter(h, 1, x) of one long stream, from
    x_local = x[plan.read_first : plan.read_first + plan.nread] (own samples + halo).  No exchange on the data path: the
    halo outputs are simply dropped.  gather=True all-gathers the pieces so that every rank holds the whole y
    (4 bytes per sample -- only for tests and small streams).  `backend(h, x)` -> filtered x; default: engine.fir_filter."""
    import torch
    import torch.distributed as dist
    if backend is None:
        from . import engine as E
        backend = lambda taps, x: E.fir_filter(taps, x)          # noqa: E731
    y = backend(h, x_local)
    own = y[plan.first - plan.read_first:]
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if world == 1 or not gather:
        return own
    t = own if isinstance(own, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(own))
    base, extra = divmod(plan.total, plan.world)
    width = base + (1 if extra else 0)
    padded = torch.zeros(width, dtype=t.dtype, device=t.device)
    padded[:t.numel()] = t
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    out = torch.cat([parts[r][:base + (1 if r < extra else 0)] for r in range(world)])
    return out if isinstance(own, torch.Tensor) else out.numpy()


def stft_sharded(x_local, win, plan, detrend=False, mean_value=None, sided=1, amp_scale=1.0, power=False, backend=None):
    """This rank's frames [plan.first_frame, plan.first_frame + plan.frames) of the STFT of one long stream (same ShardPlan
    as the PSD: contiguous frame ranges + halo).  Frames are independent: no collective; the spectrogram stays sharded
    ([plan.frames, nbins] per rank, concatenation along the frame axis is the whole result).  A global-mean detrend needs
    the mean of the whole stream: pass it as `mean_value`.  `backend(x, win, hop, frames, detrend, mean_value, sided,
    amp_scale, power)` -> [frames, nbins]; default: engine.stft_frames."""
    if backend is None:
        from . import engine as E

        def backend(x, w, hop, frames, d, mv, sd, amp, pw):
            return E.stft_frames(x, w, hop, frames, detrend=d, sided=sd, amp_scale=amp, power=pw, mean_value=mv)[0]
    return backend(x_local, win, plan.hop, plan.frames, bool(detrend) or mean_value is not None, mean_value, sided,
                   amp_scale, power)
