"""Device pipeline glue: torch owns HBM buffers and streams, libsc_hip.so does the work.

Nothing here computes on the CPU: every function launches HIP kernels / rocFFT through the
C ABI (include/sc_hip.h) on the current torch stream and returns device tensors.
"""
import ctypes
import os
from ctypes import byref, c_int64, c_void_p

import numpy as np
import torch

from . import _lib
from ._lib import SpectraDesc

# connectivity.py:67-75 of the reference: which of (window, trial, taper) are averaged
EXPECTATION_AXES = _lib.EXPECTATION_AXES

import collections

_plan_cache = collections.OrderedDict()     # (N, batch, device, f64) -> sc_fft_plan handle, least recently used first
PLAN_CACHE_SIZE = 4                          # every plan owns a rocFFT work buffer and up to 64 MB of transform scratch
_twiddle_cache = {}


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return c_void_p(t.data_ptr())


def to_host(t):
    """Device tensor -> NumPy array through a page-locked buffer (the caching host allocator keeps and reuses the
    blocks): a pageable copy of a large result runs at ~6 GB/s, a pinned one at link rate.  The array owns its
    buffer (it is the pinned tensor's memory, kept alive by the array)."""
    t = t.contiguous()
    n_bytes = t.numel() * t.element_size()
    if (1 << 20) <= n_bytes <= (2 << 30):
        try:
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            torch.cuda.current_stream(t.device).synchronize()
            return h.numpy()
        except RuntimeError:
            pass
    return t.cpu().numpy()


def fft_plan(n_fft, batch, f64=False):
    """rocFFT real-forward plan: rows [batch][N] in, frequency-major [F][batch] out.  A small LRU keeps the plans of
    the last few (N, batch, device, precision) shapes; an evicted plan is destroyed after the device has drained (its
    scratch may still be in use by queued work), so a sweep over window lengths does not pile up plan buffers."""
    key = (int(n_fft), int(batch), torch.cuda.current_device(), bool(f64))
    plan = _plan_cache.get(key)
    if plan is not None:
        _plan_cache.move_to_end(key)
        return plan
    lib = _lib.load()
    while len(_plan_cache) >= PLAN_CACHE_SIZE:
        _, old = _plan_cache.popitem(last=False)
        torch.cuda.synchronize()
        lib.sc_fft_plan_destroy(old)
    handle = c_void_p()
    create = lib.sc_fft_plan_create_f64 if f64 else lib.sc_fft_plan_create
    _lib.check(create(byref(handle), n_fft, batch), "sc_fft_plan_create")
    _plan_cache[key] = handle
    return handle


def clear_plan_cache():
    lib = _lib.load()
    if _plan_cache:
        torch.cuda.synchronize()
    for plan in _plan_cache.values():
        lib.sc_fft_plan_destroy(plan)
    _plan_cache.clear()


class DeviceSpectra:
    """One-sided (or caller-described) Fourier coefficients resident in HBM.

    ``X`` is a complex64 tensor (float32 engine) or a complex128 tensor (float64 engine, ``f64``: no pad channel,
    every consumer takes the fp64 kernels of sc_f64.hip); ``dims`` = (F, W, R, K, C) logical sizes and ``strides`` =
    element strides of (freq, window, trial, taper) -- channel stride is 1.  ``C_alloc`` >= C channels are
    stored per row: an odd channel count gets one all-zero channel appended, so that rows stay 16-byte
    aligned and the one-pass stage-B kernels (even channel counts) apply; a record accumulated over C_alloc
    channels IS the record of the first C (same 16 x 16 tiling, the extra row / column lies in tile padding).

    Planes format (float32 engine, round 4): ``P`` holds the same coefficients as two f16 pieces per real number
    (x * scale[c] = h + m, dense rows [F][W][R][K] of sc_planes_row_bytes(C) bytes: sc_fused2.hip) and ``scale`` the
    per-channel powers of two (then their reciprocals).  Stage A can write it instead of complex64 (same volume); the
    CSM / |Im s| accumulation then runs on it directly, and ``X`` is decoded from it on first use by anything else.
    """

    is_device_spectra = True        # (what Connectivity tests for: the torch-free host has a class of its own with the same mark)

    def __init__(self, X, dims, strides, n_fft, real_input, C_alloc=None, P=None, scale=None):
        self._X = X
        self.P, self.scale = P, scale
        self.F, self.W, self.R, self.K, self.C = (int(d) for d in dims)
        self.C_alloc = self.C if C_alloc is None else int(C_alloc)
        assert self.C_alloc in (self.C, self.C + 1) and -(-self.C_alloc // 16) == -(-self.C // 16)
        self.strides = tuple(int(s) for s in strides)
        self.n_fft = int(n_fft)
        self.real_input = bool(real_input)   # negative bins are conj mirrors of positive ones
        self.f64 = X is not None and X.dtype == torch.complex128
        # planes format written by stage A: ``quality`` is a device scalar, min over the channels of (typical sample magnitude x
        # channel scale); times ``taper_l2_min`` it is the typical coefficient in scaled units, which the caller that owns the series
        # compares with _lib.PLANES_MIN_TYPICAL (Multitaper.device_spectra does; planes_typical_coefficient() reads it back)
        self.quality = self.taper_l2_min = None
        self.device = X.device if X is not None else P.device

    def planes_typical_coefficient(self):
        """Smallest typical coefficient over the channels, in the scaled units of the f16 pieces (one small read-back)."""
        return float(self.quality.item()) * self.taper_l2_min

    @property
    def X(self):
        """The complex64 / complex128 coefficients; decoded from the planes format (lossless up to its 22 bits) on first use."""
        if self._X is None:
            lib = _lib.load()
            X = torch.empty((self.F, self.W, self.R, self.K, self.C_alloc), dtype=torch.complex64, device=self.P.device)
            d = self.desc("trials_tapers", padded=True)
            _lib.check(lib.sc_spectra_from_planes_f32(_ptr(self.P), byref(d), _ptr(self.scale), _ptr(X), _stream()),
                       "sc_spectra_from_planes_f32")
            self._X = X
        return self._X

    def coefficients(self):
        """The spectra as a (F, W, R, K, C) tensor view of a contiguous X (the zero pad channel dropped)."""
        return self.X.view(self.F, self.W, self.R, self.K, self.C_alloc)[..., :self.C]

    def freq_slice(self, f0, f1):
        """The bins [f0, f1) as a view (no copy): same strides, pointer advanced by f0 * stride_freq."""
        assert 0 <= f0 < f1 <= self.F
        P = flat = None
        if self.P is not None:
            per_bin = self.W * self.R * self.K * int(_lib.load().sc_planes_row_bytes(self.C_alloc))
            P = self.P[f0 * per_bin:f1 * per_bin]
        if self._X is not None:
            assert self._X.is_contiguous()
            flat = self._X.view(-1)[f0 * self.strides[0]:]
        return DeviceSpectra(flat, (f1 - f0, self.W, self.R, self.K, self.C), self.strides, self.n_fft,
                             self.real_input, C_alloc=self.C_alloc, P=P, scale=self.scale)

    def desc(self, expectation_type, n_freq=None, padded=False):
        """Descriptor of the spectra; ``padded``: with the zero pad channel counted as a signal."""
        axes = EXPECTATION_AXES[expectation_type]
        sF, sW, sR, sK = self.strides
        return SpectraDesc(n_freq=self.F if n_freq is None else n_freq, n_windows=self.W,
                           n_trials=self.R, n_tapers=self.K, n_signals=self.C_alloc if padded else self.C, stride_freq=sF,
                           stride_window=sW, stride_trial=sR, stride_taper=sK,
                           reduce_window=int(0 in axes), reduce_trial=int(1 in axes),
                           reduce_taper=int(2 in axes), reserved=0)


def _taper_norms(tapers_over_fs):
    """(max_k sum_n |h_k[n]|, min_k ||h_k||_2): the bound behind the channel scales of the planes format and the size of a typical
    coefficient per unit of sample spread -- properties of the tapers, kept ON the tensor object together with its version
    counter: one device synchronisation per taper tensor, not per transform."""
    cached = getattr(tapers_over_fs, "_sc_norms", None)
    if cached is None or cached[1] != tapers_over_fs._version:
        both = torch.stack([tapers_over_fs.abs().sum(dim=1).max(), tapers_over_fs.pow(2).sum(dim=1).sqrt().min()]).tolist()
        cached = ((float(both[0]), float(both[1])), tapers_over_fs._version)
        tapers_over_fs._sc_norms = cached
    return cached[0]


def twiddles(n_fft, device):
    """exp(-2 pi i m / N) table for the fused FFT kernel (device, cached per (N, device))."""
    key = (int(n_fft), str(device))
    tw = _twiddle_cache.get(key)
    if tw is None:
        lib = _lib.load()
        tw = torch.empty((n_fft,), dtype=torch.complex64, device=device)
        _lib.check(lib.sc_fft_twiddles_f32(n_fft, _ptr(tw), _stream()), "sc_fft_twiddles_f32")
        _twiddle_cache[key] = tw
    return tw


PLANES_FORMAT_FAMILIES = _lib.PLANES_FORMAT_FAMILIES
planes_format_applies = _lib.planes_format_applies


def multitaper_spectra(x, tapers_over_fs, n_window, n_step, n_fft, n_windows, detrend_type, mark=None,
                       use_fused=None, n_signals=None, planes_hint=None):
    """Stage A on device: (T,R,C) float32 tensor -> DeviceSpectra [F][W][R][K][C].

    ``tapers_over_fs``: (K, L) float32 device tensor = reference tapers^T / fs
    (folds the sqrt(fs) of transforms.py:1440 and the /fs of transforms.py:1405).
    ``n_signals``: number of real channels when ``x`` already carries the all-zero pad channel of an odd channel count
    (appended on the host before the upload, transforms.Multitaper.device_spectra); a device tensor with an odd channel
    count that arrives unpadded is copied into a padded buffer here (one strided device copy).
    ``planes_hint``: the accumulator families the caller will ask for.  Any family sc_fused2.hip serves, 44 ... 1024 signals, a
    window length stage A has the output for (the powers of two 64 ... 4096, the lengths 200 ... 2000 of sc_mtfft_mixed.hip:
    sc_multitaper_fft_planes_supported) and at least 256 MB of spectra (_lib.planes_format_applies): the spectra are
    written in the planes format (two f16 pieces per real number) -- a scan of the series for the channel scales, then the same
    fused transform.  The scan also reports how large a typical coefficient will be in the format's scaled units
    (``DeviceSpectra.planes_typical_coefficient()``): one scale per channel serves every window, so the format is meant for
    series without samples hundreds of times the typical amplitude; Multitaper.device_spectra checks against
    _lib.PLANES_MIN_TYPICAL and re-runs the transform into complex64 otherwise.
    """
    lib = _lib.load()
    T, R, C_real = x.shape
    if n_signals is not None:
        assert n_signals in (C_real, C_real - 1)
        C_real = int(n_signals)
    elif C_real % 2 and C_real + 1 <= _lib.PLANES_FORMAT_MAX_CHANNELS:
        padded = torch.zeros((T, R, C_real + 1), dtype=x.dtype, device=x.device)
        padded[..., :C_real].copy_(x)                # odd channel count: one zero channel (see DeviceSpectra)
        x = padded
    T, R, C = x.shape
    K, L = tapers_over_fs.shape
    assert L == n_window
    F = n_fft // 2 + 1
    strides = (n_windows * R * K * C, R * K * C, K * C, C)
    if use_fused is None:
        use_fused = bool(lib.sc_multitaper_fft_supported(L, n_fft))
    if use_fused and planes_format_applies(L, n_fft, C, planes_hint, spectra_bytes=F * n_windows * R * K * C * 8):
        row_bytes = int(lib.sc_planes_row_bytes(C))
        P = torch.empty((F * n_windows * R * K * row_bytes,), dtype=torch.uint8, device=x.device)
        scale = torch.empty((2 * C,), dtype=torch.float32, device=x.device)
        work_bytes = int(lib.sc_planes_scales_work_bytes(T * R, C))
        work = torch.empty((work_bytes,), dtype=torch.uint8, device=x.device)
        quality = torch.empty((1,), dtype=torch.float32, device=x.device)      # DeviceSpectra.quality, see there
        abs_sum, l2_min = _taper_norms(tapers_over_fs)
        _lib.check(lib.sc_planes_scales_quality_f32(_ptr(x), T, R, C, _lib.DETREND[detrend_type], abs_sum, _ptr(scale), _ptr(work),
                                                    work_bytes, _ptr(quality), _stream()), "sc_planes_scales_quality_f32")
        _lib.check(lib.sc_multitaper_fft_planes_f32(_ptr(x), T, R, C, L, n_step, n_windows, n_fft, _ptr(tapers_over_fs), K,
                                                    _lib.DETREND[detrend_type], _ptr(twiddles(n_fft, x.device)), _ptr(scale),
                                                    _ptr(P), _stream()), "sc_multitaper_fft_planes_f32")
        if mark:
            mark("mtfft_fused")
        sp = DeviceSpectra(None, (F, n_windows, R, K, C_real), strides, n_fft, real_input=True, C_alloc=C, P=P, scale=scale)
        sp.quality, sp.taper_l2_min = quality, l2_min
        return sp
    if use_fused:
        # one kernel: window + detrend + taper + FFT + transposed store (sc_mtfft.hip)
        X = torch.empty((F, n_windows, R, K, C), dtype=torch.complex64, device=x.device)
        _lib.check(lib.sc_multitaper_fft_f32(_ptr(x), T, R, C, L, n_step, n_windows, n_fft,
                                             _ptr(tapers_over_fs), K, _lib.DETREND[detrend_type],
                                             _ptr(twiddles(n_fft, x.device)), _ptr(X), _stream()),
                   "sc_multitaper_fft_f32")
        if mark:
            mark("mtfft_fused")
        return DeviceSpectra(X, (F, n_windows, R, K, C_real), strides, n_fft, real_input=True, C_alloc=C)
    batch = n_windows * R * K * C
    y = torch.empty((batch, n_fft), dtype=torch.float32, device=x.device)
    _lib.check(lib.sc_taper_windows_f32(_ptr(x), T, R, C, L, n_step, n_windows, n_fft,
                                        _ptr(tapers_over_fs), K, _lib.DETREND[detrend_type],
                                        _ptr(y), _stream()), "sc_taper_windows_f32")
    if mark:
        mark("taper_windows")
    X = torch.empty((F, n_windows, R, K, C), dtype=torch.complex64, device=x.device)
    _lib.check(lib.sc_fft_execute(fft_plan(n_fft, batch), _ptr(y), _ptr(X), _stream()), "sc_fft_execute")
    if mark:
        mark("rocfft_r2c")
    del y
    return DeviceSpectra(X, (F, n_windows, R, K, C_real), strides, n_fft, real_input=True, C_alloc=C)


def multitaper_spectra_f64(x, tapers_over_fs, n_window, n_step, n_fft, n_windows, detrend_type, mark=None, use_fused=None):
    """Stage A of the float64 engine: (T,R,C) float64 tensor -> complex128 DeviceSpectra [F][W][R][K][C].  One fused kernel
    (sc_multitaper_fft_f64) for the lengths it compiles; sc_taper_windows_f64 + double-precision rocFFT + transpose for
    any other window / FFT length."""
    lib = _lib.load()
    T, R, C = x.shape
    K, L = tapers_over_fs.shape
    assert L == n_window and x.dtype == torch.float64 and tapers_over_fs.dtype == torch.float64
    F = n_fft // 2 + 1
    strides = (n_windows * R * K * C, R * K * C, K * C, C)
    batch = n_windows * R * K * C
    if use_fused is None:
        use_fused = bool(lib.sc_multitaper_fft_f64_supported(L, n_fft)) and R <= 65535 and n_windows <= 65535
    if use_fused:
        X = torch.empty((F, n_windows, R, K, C), dtype=torch.complex128, device=x.device)
        _lib.check(lib.sc_multitaper_fft_f64(_ptr(x), T, R, C, L, n_step, n_windows, n_fft, _ptr(tapers_over_fs), K,
                                             _lib.DETREND[detrend_type], _ptr(X), _stream()), "sc_multitaper_fft_f64")
        if mark:
            mark("mtfft_fused_f64")
        return DeviceSpectra(X, (F, n_windows, R, K, C), strides, n_fft, real_input=True)
    y = torch.empty((batch, n_fft), dtype=torch.float64, device=x.device)
    _lib.check(lib.sc_taper_windows_f64(_ptr(x), T, R, C, L, n_step, n_windows, n_fft, _ptr(tapers_over_fs), K,
                                        _lib.DETREND[detrend_type], _ptr(y), _stream()), "sc_taper_windows_f64")
    if mark:
        mark("taper_windows_f64")
    X = torch.empty((F, n_windows, R, K, C), dtype=torch.complex128, device=x.device)
    _lib.check(lib.sc_fft_execute_f64(fft_plan(n_fft, batch, f64=True), _ptr(y), _ptr(X), _stream()),
               "sc_fft_execute_f64")
    if mark:
        mark("rocfft_d2z")
    del y
    return DeviceSpectra(X, (F, n_windows, R, K, C), strides, n_fft, real_input=True)


def upload_coefficients(coef, device="cuda", f64=False):
    """Reference-layout (W,R,K,N,C) complex coefficients -> DeviceSpectra (all N bins, as given)."""
    coef = np.asarray(coef)
    W, R, K, N, C_real = coef.shape
    if f64:
        X = torch.from_numpy(np.ascontiguousarray(coef, dtype=np.complex128)).to(device)
        return DeviceSpectra(X, (N, W, R, K, C_real), (C_real, R * K * N * C_real, K * N * C_real, N * C_real), N,
                             real_input=False)
    coef = np.ascontiguousarray(coef, dtype=np.complex64)
    if C_real % 2 and C_real + 1 <= 256:
        coef = np.concatenate([coef, np.zeros(coef.shape[:-1] + (1,), dtype=np.complex64)], axis=-1)
    C = coef.shape[-1]
    X = torch.from_numpy(coef).to(device)
    return DeviceSpectra(X, (N, W, R, K, C_real), (C, R * K * N * C, K * N * C, N * C), N, real_input=False, C_alloc=C)


def accum_layout(spectra, expectation_type, planes, n_freq=None):
    lib = _lib.load()
    d = spectra.desc(expectation_type, n_freq)
    n_bins, fpb, n_groups, n_obs = c_int64(), c_int64(), c_int64(), c_int64()
    _lib.check(lib.sc_accum_layout(byref(d), planes, byref(n_bins), byref(fpb), byref(n_groups),
                                   byref(n_obs)), "sc_accum_layout")
    return n_bins.value, fpb.value, n_groups.value, n_obs.value


_ws_cache = {}


def _workspace(n_bytes, device, owner=None):
    """Scratch the fused stage-B kernel uses to split bins over workgroups (kept and reused per device).  ``owner``: a dict of
    the caller's in which the buffer lives instead of the per-device cache -- a captured pass (GraphedMeasures) replays the
    buffer's ADDRESS, so it must not be the shared one, which a later, larger request replaces and frees."""
    if n_bytes <= 0:
        return None
    cache = _ws_cache if owner is None else owner
    key = (device.type, device.index)
    buf = cache.get(key)
    if buf is None or buf.numel() < n_bytes:
        buf = torch.empty(n_bytes, dtype=torch.uint8, device=device)
        cache[key] = buf
    return buf


def _record_tensor(n_bins, fpb, dtype, device, row_multiple):
    """[n_bins, fpb] records; with row_multiple > 1 the allocation is rounded up to that many rows (zeroed tail) and
    the first n_bins rows are returned as a view of it -- the padded buffer (``._base``) is what a reduce-scatter over
    bins takes, without a concatenation."""
    rows = -(-n_bins // row_multiple) * row_multiple
    full = torch.empty((rows, fpb), dtype=dtype, device=device)
    if rows > n_bins:
        full[n_bins:].zero_()
    return full[:n_bins]


_PLANE_WIDTH = ((_lib.PLANE_CSM, 2), (_lib.PLANE_ABS_IM, 1), (_lib.PLANE_IM_SQ, 1), (_lib.PLANE_SIGN_IM, 1),
                (_lib.PLANE_UNIT, 2))          # record order and planes per family (sc_common.h: sc_plane_offset)


def plane_slots(planes):
    """{family bit: (first plane index, n planes)} of a record with the families ``planes``."""
    out, n = {}, 0
    for bit, width in _PLANE_WIDTH:
        if planes & bit:
            out[bit] = (n, width)
            n += width
    return out


MAX_KERNEL_SIGNALS = 256          # SC_MAX_SIGNALS of csrc/sc_common.h: what one launch of the stage-B kernels stages per observation row
BLOCK_SIGNALS = 128               # channel block of the tiling beyond that (a multiple of the 16-channel record tile)


def _channel_subset(spectra, cols):
    """The spectra of the channels ``cols`` (a LongTensor of channel indices) as a dense DeviceSpectra of its own: one gathering
    copy; an odd count gets the zero pad channel of the float32 engine."""
    X = spectra.X
    n = int(cols.numel())
    n_alloc = n if (spectra.f64 or n % 2 == 0) else n + 1
    sub = torch.zeros(tuple(X.shape[:-1]) + (n_alloc,), dtype=X.dtype, device=X.device) if n_alloc != n else \
        torch.empty(tuple(X.shape[:-1]) + (n_alloc,), dtype=X.dtype, device=X.device)
    torch.index_select(X, X.dim() - 1, cols, out=sub[..., :n]) if n_alloc == n else sub[..., :n].copy_(X.index_select(X.dim() - 1, cols))
    assert all(st % spectra.C_alloc == 0 for st in spectra.strides), "channel subsets need spectra whose rows are dense"
    strides = tuple(st // spectra.C_alloc * n_alloc for st in spectra.strides)
    return DeviceSpectra(sub, (spectra.F, spectra.W, spectra.R, spectra.K, n), strides, spectra.n_fft, spectra.real_input, C_alloc=n_alloc)


def _accumulate_blocked(spectra, expectation_type, planes, n_freq, mark, row_multiple):
    """Stage B for MORE signals than one launch of the kernels stages (256): the reference has no limit
    (connectivity.py:447-526), a 306-channel MEG array is an ordinary input.  The channels are cut into blocks of 128; every pair
    of blocks (a < b) is accumulated as a request of its own on the gathered spectra of the two blocks (<= 256 signals: the
    ordinary kernels), and its 16 x 16 record tiles are copied to their places in the full record -- the cross tiles of (a, b)
    from that pair, the tiles inside block a from the pair (a, a + 1) (the last block from the last pair).  Every entry of the
    record is computed by the same kernels as for <= 256 signals; what the tiling costs is the tiles inside the blocks being
    computed once per partner (about twice the arithmetic of an untiled triangle at three blocks) and one gathering copy of the
    spectra per pair."""
    C = spectra.C
    n_blk = -(-C // BLOCK_SIGNALS)
    n_bins, fpb, _, n_obs = accum_layout(spectra, expectation_type, planes, n_freq)
    NB = -(-C // 16)
    n_tiles = NB * (NB + 1) // 2
    n_planes = fpb // (n_tiles * 256)
    dtype = torch.float64 if spectra.f64 else torch.float32
    full = _record_tensor(n_bins, fpb, dtype, spectra.device, row_multiple)
    full_v = full.view(n_bins, n_planes, n_tiles, 256)
    dev = spectra.device

    def tile(bi, bj, nb):
        return bi * nb - bi * (bi - 1) // 2 + (bj - bi)

    per = BLOCK_SIGNALS // 16
    for a in range(n_blk - 1):
        for b in range(a + 1, n_blk):
            ca = torch.arange(a * BLOCK_SIGNALS, (a + 1) * BLOCK_SIGNALS, device=dev)
            cb = torch.arange(b * BLOCK_SIGNALS, min((b + 1) * BLOCK_SIGNALS, C), device=dev)
            sub = _channel_subset(spectra, torch.cat([ca, cb]))
            rec, _ = accumulate(sub, expectation_type, planes, n_freq=n_freq, mark=mark)
            nb_s = -(-sub.C // 16)
            rec_v = rec.view(n_bins, n_planes, nb_s * (nb_s + 1) // 2, 256)
            src, dst = [], []
            for ti in range(nb_s):
                for tj in range(ti, nb_s):
                    in_a_i, in_a_j = ti < per, tj < per
                    if in_a_i and in_a_j:
                        keep = b == a + 1                                   # inside block a: from its first partner
                    elif not in_a_i and not in_a_j:
                        keep = a == n_blk - 2 and b == n_blk - 1            # inside the last block: from the last pair
                    else:
                        keep = True                                         # cross tiles of (a, b)
                    if keep:
                        gi = a * per + ti if in_a_i else b * per + (ti - per)
                        gj = a * per + tj if in_a_j else b * per + (tj - per)
                        src.append(tile(ti, tj, nb_s))
                        dst.append(tile(gi, gj, NB))
            src_t, dst_t = torch.tensor(src, device=dev), torch.tensor(dst, device=dev)
            full_v[:, :, dst_t] = rec_v[:, :, src_t]
            del rec, sub
    return full, n_obs


def accumulate(spectra, expectation_type, planes, n_freq=None, mark=None, use_fused=None, row_multiple=1, have=None,
               fold=True, ws_owner=None):
    """Stage B: un-normalised accumulator record tensor [n_bins, floats_per_bin] (float32; float64 records from
    complex128 spectra).  ``row_multiple``: see _record_tensor (trial-sharded callers pass the world size).
    ``fold=False`` (planes-format path only; ignored elsewhere): when stage B split every bin over several workgroups, their
    partial records are NOT summed -- the result is then ONE 3-D tensor [n_parts, n_bins, floats_per_bin] of its own (the sum
    over axis 0, in part order, is the record): measure() / measure_multi() add the parts while their kernel reads them (one pass
    and one record round trip less), fold_parts() gives the 2-D record to any other consumer.
    ``have`` = (planes_old, record_old), float64 engine only: families already accumulated for the same spectra and
    expectation are copied over (a strided device copy) and only the missing ones are computed -- its CSM and
    per-observation planes are separate kernels, so a wPLI after a coherence costs the |Im s| plane alone.
    ``ws_owner``: see _workspace (a dict that owns the split-bin scratch of this call instead of the per-device cache)."""
    lib = _lib.load()
    if spectra.C > MAX_KERNEL_SIGNALS:
        # planes-format spectra go straight to sc_fused2.hip, which plans its launches over any number of 32-channel blocks (round 6);
        # every other request beyond 256 signals is tiled over channel-block pairs
        direct = (not spectra.f64 and spectra.P is not None and use_fused is not False
                  and bool(lib.sc_fused2_supported(byref(spectra.desc(expectation_type, n_freq, padded=True)), planes)))
        if not direct:
            return _accumulate_blocked(spectra, expectation_type, planes, n_freq, mark, row_multiple)
    d = spectra.desc(expectation_type, n_freq)
    n_bins, fpb, _, n_obs = accum_layout(spectra, expectation_type, planes, n_freq)
    if spectra.f64:
        # float64 engine: fp64 matrix cores for the CSM planes, fp64 VALU for the others, double records
        accum = _record_tensor(n_bins, fpb, torch.float64, spectra.device, row_multiple)
        which = planes
        if have is not None and have[1].dtype == torch.float64 and have[1].shape[0] == n_bins and (have[0] & planes):
            old_planes, old = have
            new_slots, old_slots = plane_slots(planes), plane_slots(old_planes)
            n_new, n_old = sum(w for _, w in new_slots.values()), sum(w for _, w in old_slots.values())
            new_v, old_v = accum.view(n_bins, n_new, fpb // n_new), old.view(n_bins, n_old, old.shape[1] // n_old)
            for bit, (i_new, width) in new_slots.items():
                if bit in old_slots:
                    new_v[:, i_new:i_new + width].copy_(old_v[:, old_slots[bit][0]:old_slots[bit][0] + width])
                    which &= ~bit
        if which:
            _lib.check(lib.sc_accumulate_f64(_ptr(spectra.X), byref(d), planes, which, _ptr(accum), _stream()),
                       "sc_accumulate_f64")
        if mark:
            mark("accumulate_f64")
        return accum, n_obs
    accum = None
    if spectra.P is not None and use_fused is not False:
        dp = spectra.desc(expectation_type, n_freq, padded=True)
        if lib.sc_fused2_supported(byref(dp), planes):
            # planes format: CSM (+ |Im s|) straight from the f16 pieces stage A wrote (sc_fused2.hip)
            ws_bytes = int(lib.sc_fused_workspace_bytes(byref(dp), planes))
            part_bytes = n_bins * fpb * 4
            if not fold and ws_bytes >= part_bytes and row_multiple == 1:
                # partial records kept: parts 1 .. behind part 0 in one allocation of the caller's own
                max_parts = 1 + ws_bytes // part_bytes
                parts = torch.empty((max_parts, n_bins, fpb), dtype=torch.float32, device=spectra.device)
                n_parts = ctypes.c_int(1)
                _lib.check(lib.sc_fused2_csm_absim_parts_f32(_ptr(spectra.P), byref(dp), _ptr(spectra.scale), planes, _ptr(parts[0]),
                                                             _ptr(parts[1]), (max_parts - 1) * part_bytes, byref(n_parts), _stream()),
                           "sc_fused2_csm_absim_parts_f32")
                if mark:
                    mark("fused2_csm_absim")
                return (parts[:n_parts.value] if n_parts.value > 1 else parts[0]), n_obs
            ws = _workspace(ws_bytes, spectra.device, ws_owner)
            accum = _record_tensor(n_bins, fpb, torch.float32, spectra.device, row_multiple)
            _lib.check(lib.sc_fused2_csm_absim_f32(_ptr(spectra.P), byref(dp), _ptr(spectra.scale), planes, _ptr(accum),
                                                   _ptr(ws) if ws is not None else None, ws_bytes, _stream()),
                       "sc_fused2_csm_absim_f32")
            if mark:
                mark("fused2_csm_absim")
            return accum, n_obs
    accum = _record_tensor(n_bins, fpb, torch.float32, spectra.device, row_multiple)
    per_plane_only = use_fused is False        # explicit request (tests): every plane through its separate kernel
    if use_fused is None:
        use_fused = bool(lib.sc_fused_supported(spectra.C_alloc))
    # planes the one-pass kernels fill for this shape (sc_fused.hip): CSM, |Im s|, s/|s|; for few channels also
    # (Im s)^2 and sign(Im s).  Whatever is left goes to the per-plane VALU kernel.  The one-pass kernels see the zero
    # pad channel of an odd channel count as a signal: same record (DeviceSpectra).
    d_real, d = d, spectra.desc(expectation_type, n_freq, padded=True)
    one_pass = int(lib.sc_fused_planes_covered(byref(d), planes)) if use_fused else 0
    if one_pass:
        ws_bytes = int(lib.sc_fused_workspace_bytes(byref(d), planes))
        ws = _workspace(ws_bytes, spectra.device, ws_owner)
        ws_ptr = _ptr(ws) if ws is not None else None
        if one_pass & _lib.PLANE_CSM:
            # CSM (+ the per-observation |Im s| products, + (Im s)^2): bf16 matrix pipe, or the f32 VALU kernel
            _lib.check(lib.sc_fused_csm_absim_ws_f32(_ptr(spectra.X), byref(d), planes, _ptr(accum), ws_ptr, ws_bytes,
                                                     _stream()), "sc_fused_csm_absim_ws_f32")
            if mark:
                mark("fused_csm_absim")
        if one_pass & _lib.PLANE_SIGN_IM:
            _lib.check(lib.sc_fused_sign_ws_f32(_ptr(spectra.X), byref(d), planes, _ptr(accum), ws_ptr, ws_bytes,
                                                _stream()), "sc_fused_sign_ws_f32")
            if mark:
                mark("fused_sign")
        if one_pass & _lib.PLANE_UNIT:
            # sum s/|s| = the CSM of the unit phasors x/|x|: the same kernels on normalised rows
            sb = int(lib.sc_fused_unit_scratch_bytes(byref(d)))
            scratch = torch.empty((sb,), dtype=torch.uint8, device=spectra.device) if sb else None
            _lib.check(lib.sc_fused_unit_ws_f32(_ptr(spectra.X), byref(d), planes, _ptr(accum), ws_ptr, ws_bytes,
                                                _ptr(scratch) if scratch is not None else None, sb, _stream()),
                       "sc_fused_unit_ws_f32")
            if mark:
                mark("fused_unit")
        nl = planes & ~one_pass
        if nl:
            _lib.check(lib.sc_nonlinear_accumulate_f32(_ptr(spectra.X), byref(d_real), planes, nl, _ptr(accum),
                                                       _stream()), "sc_nonlinear_accumulate_f32")
            if mark:
                mark("nonlinear_valu")
        return accum, n_obs
    d = d_real
    if planes & _lib.PLANE_CSM:
        _lib.check(lib.sc_csm_accumulate_f32(_ptr(spectra.X), byref(d), planes, _ptr(accum), _stream()),
                   "sc_csm_accumulate_f32")
        if mark:
            mark("csm_mfma")
    nl = planes & ~_lib.PLANE_CSM
    if nl & _lib.PLANE_UNIT and not per_plane_only:
        # sum s/|s| as the CSM of a normalised copy of the spectra (f32 MFMA) instead of a per-pair rsqrt on the VALU
        sb = int(lib.sc_unit_scratch_bytes(byref(d)))
        scratch = torch.empty((sb,), dtype=torch.uint8, device=spectra.device)
        _lib.check(lib.sc_unit_accumulate_f32(_ptr(spectra.X), byref(d), planes, _ptr(accum), _ptr(scratch), sb,
                                              _stream()), "sc_unit_accumulate_f32")
        nl &= ~_lib.PLANE_UNIT
        if mark:
            mark("unit_mfma")
    if nl:
        _lib.check(lib.sc_nonlinear_accumulate_f32(_ptr(spectra.X), byref(d), planes, nl, _ptr(accum),
                                                   _stream()), "sc_nonlinear_accumulate_f32")
        if mark:
            mark("nonlinear_valu")
    return accum, n_obs


def rec_planes(accum, planes):
    """`planes` as the consumers of a record tensor want it: with SC_RECORD_F64 when the records are doubles."""
    return (planes | _lib.RECORD_F64) if accum.dtype == torch.float64 else (planes & ~_lib.RECORD_F64)


def fold_parts(accum):
    """[n_parts, n_bins, floats_per_bin] partial records -> their sum in part order (the record a folding pass of stage B
    would have written, bit for bit: the same additions in the same order); a 2-D record is returned as it is.  The folded
    record is kept on the tensor, so several consumers pay for one fold."""
    if accum.dim() != 3:
        return accum
    cached = getattr(accum, "_sc_folded", None)
    if cached is None:
        cached = accum[0].clone()
        for k in range(1, accum.shape[0]):
            cached.add_(accum[k])
        accum._sc_folded = cached
    return cached


def measure(accum, n_signals, planes, n_obs, which, out=None, wide=None):
    """Stage C: one measure from an accumulator tensor (after any cross-GPU sum).  ``wide``: write float64 /
    complex128 (what the reference returns) straight from the epilogue; default: wide for double records."""
    lib = _lib.load()
    parts = None
    if accum.dim() == 3:
        # partial records (accumulate(fold=False), or the blocks of a direct exchange): the epilogue kernels sum them in part order
        # while they read (sc_measure_parts: every measure, power and the complex-valued ones included)
        if accum.shape[0] > 1 and accum.is_contiguous():
            parts, accum = accum, accum[0]
        else:
            accum = fold_parts(accum)
    n_bins = accum.shape[0]
    C = n_signals
    if wide is None:
        wide = accum.dtype == torch.float64
    real_t, cplx_t = (torch.float64, torch.complex128) if wide else (torch.float32, torch.complex64)
    if which == _lib.M_POWER:
        shape, dtype = (n_bins, C), real_t
    elif which in _lib.COMPLEX_MEASURES:
        shape, dtype = (n_bins, C, C), cplx_t
    else:
        shape, dtype = (n_bins, C, C), real_t
    if out is None:
        out = torch.empty(shape, dtype=dtype, device=accum.device)
    if parts is not None:
        _lib.check(lib.sc_measure_parts(_ptr(parts[0]), _ptr(parts[1]), parts.shape[0], parts.stride(0), n_bins, C,
                                        rec_planes(accum, planes), n_obs, which, _ptr(out), int(bool(wide)), _stream()), "sc_measure_parts")
        return out
    fn = lib.sc_measure_f64 if wide else lib.sc_measure_f32
    _lib.check(fn(_ptr(accum), n_bins, C, rec_planes(accum, planes), n_obs, which, _ptr(out), _stream()),
               "sc_measure")
    return out


MEASURE_MULTI_MAX = 4


def measure_multi(accum, n_signals, planes, n_obs, which, wide=None, stacked=False):
    """Stage C for several real-valued C x C measures of one record: ONE launch reads the record once
    (sc_measure_multi_*); complex measures / power, or more than four, go through measure().
    ``stacked``: the results are the slices of ONE [n_measures, n_bins, C, C] tensor (returned as ``outs[0]._base``'s
    views) when the one-launch form applies -- the trial-sharded path then gathers all measures in one collective.
    ``accum`` may be 3-D, [n_parts, n_bins, floats_per_bin]: partial records (the blocks received from the other ranks)
    that the epilogue sums in part order while it reads them (sc_measure_multi_parts) -- or, where the one-launch form
    does not apply, that are summed first."""
    which = list(which)
    parts = None
    simple = [w for w in which if w != _lib.M_POWER and w not in _lib.COMPLEX_MEASURES]
    if accum.dim() == 3:
        if accum.shape[0] > 1 and len(simple) == len(which) and 1 <= len(which) <= MEASURE_MULTI_MAX and accum.is_contiguous():
            parts = accum
            accum = parts[0]
        else:
            accum = fold_parts(accum)                                # part (= rank) order
    if parts is None and (len(simple) != len(which) or not 2 <= len(which) <= MEASURE_MULTI_MAX):
        return [measure(accum, n_signals, planes, n_obs, w, wide=wide) for w in which]
    lib = _lib.load()
    n_bins, C = accum.shape[0], n_signals
    if wide is None:
        wide = accum.dtype == torch.float64
    if stacked:
        block = torch.empty((len(which), n_bins, C, C), dtype=torch.float64 if wide else torch.float32, device=accum.device)
        outs = list(block.unbind(0))
    else:
        outs = [torch.empty((n_bins, C, C), dtype=torch.float64 if wide else torch.float32, device=accum.device) for _ in which]
    ids = (ctypes.c_int * len(which))(*which)
    ptrs = (ctypes.c_void_p * len(which))(*[o.data_ptr() for o in outs])
    if parts is not None:
        _lib.check(lib.sc_measure_multi_parts(_ptr(parts[0]), _ptr(parts[1]), parts.shape[0], parts.stride(0), n_bins, C,
                                              rec_planes(accum, planes), n_obs, len(which), ids, ptrs, int(bool(wide)), _stream()),
                   "sc_measure_multi_parts")
        return outs
    fn = lib.sc_measure_multi_f64 if wide else lib.sc_measure_multi_f32
    _lib.check(fn(_ptr(accum), n_bins, C, rec_planes(accum, planes), n_obs, len(which), ids, ptrs, _stream()),
               "sc_measure_multi")
    return outs


MAX_WILSON_ITERATIONS = 1024      # iterations the device kernels can log (WILSON_HIST / MV_HIST in csrc)


def check_max_iterations(max_iterations):
    """The reference takes any positive count (minimum_phase_decomposition.py:227-322); the device kernels log at most 1024."""
    if not 1 <= int(max_iterations) <= MAX_WILSON_ITERATIONS:
        raise ValueError(f"max_iterations must be between 1 and {MAX_WILSON_ITERATIONS} on the device path (got {max_iterations}); "
                         "Wilson's iteration converges in tens of steps or not at all")
    return int(max_iterations)


GRANGER_WORK_BYTES = 8 << 30      # workspace bound of one sc_granger_pairwise_f64 call (160 bytes per problem and bin)


def granger_pairwise(accum, n_groups, n_freq_accum, n_fft, n_signals, planes, n_obs, pairs,
                     tolerance=1e-8, max_iterations=60):
    """Batched 2x2 Wilson + spectral Granger (sc_wilson.hip).  Returns (out, n_iter, status, summary) with
    summary = (iterations run, problems not converged, problems started from the identity because their lag-0
    covariance was not positive definite).  A long pair list is walked in chunks that bound the workspace; every
    chunk writes its pairs into the same output."""
    lib = _lib.load()
    max_iterations = check_max_iterations(max_iterations)
    dev = accum.device
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    n_pairs = pairs.shape[0]
    F = n_fft // 2 + 1
    out = torch.empty((n_groups, F, n_signals, n_signals), dtype=torch.float64, device=dev)
    n_iter = torch.empty((n_groups, n_pairs), dtype=torch.int32, device=dev)
    status = torch.empty((n_groups, n_pairs), dtype=torch.int32, device=dev)
    per_pair = n_groups * n_fft * 160
    chunk = int(max(1, min(n_pairs, GRANGER_WORK_BYTES // per_pair)))
    nbytes = ctypes.c_size_t()
    _lib.check(lib.sc_granger_workspace_bytes(n_groups, chunk, n_fft, byref(nbytes)), "sc_granger_workspace_bytes")
    work = torch.empty((nbytes.value,), dtype=torch.uint8, device=dev)
    iters = not_conv = fallback = 0
    for p0 in range(0, n_pairs, chunk):
        n = min(chunk, n_pairs - p0)
        pairs_t = torch.from_numpy(pairs[p0:p0 + n]).to(dev)
        it_c = torch.empty((n_groups * n,), dtype=torch.int32, device=dev)
        st_c = torch.empty((n_groups * n,), dtype=torch.int32, device=dev)
        summary = (ctypes.c_int32 * 3)(0, 0, 0)
        _lib.check(lib.sc_granger_pairwise_f64(_ptr(accum), n_groups, n_freq_accum, n_fft, n_signals,
                                               rec_planes(accum, planes), n_obs,
                                               _ptr(pairs_t), n, tolerance, max_iterations, _ptr(work), nbytes.value,
                                               _lib.GRANGER_KEEP_OUTPUT if p0 else 0, _ptr(out), _ptr(it_c),
                                               _ptr(st_c), summary, _stream()), "sc_granger_pairwise_f64")
        n_iter[:, p0:p0 + n] = it_c.view(n_groups, n)
        status[:, p0:p0 + n] = st_c.view(n_groups, n)
        iters = max(iters, summary[0])
        not_conv += summary[1]
        fallback += summary[2]
    return out, n_iter.reshape(-1), status.reshape(-1), (iters, not_conv, fallback)


def _mvar_workspace(n_groups, n_signals, n_fft, dev):
    lib = _lib.load()
    nbytes = ctypes.c_size_t()
    _lib.check(lib.sc_mvar_workspace_bytes(n_groups, n_signals, n_fft, byref(nbytes)), "sc_mvar_workspace_bytes")
    return torch.empty((nbytes.value,), dtype=torch.uint8, device=dev), nbytes.value


def mvar_factor(n_groups, n_fft, n_signals, accum=None, n_freq_accum=0, planes=0, n_obs=1, spectra=None,
                tolerance=1e-8, max_iterations=60):
    """Full C x C Wilson factor (sc_mvar.hip) of accumulator records or of a two-sided complex128 spectrum
    tensor [n_groups, n_fft, C, C].  Returns (G [n_groups, n_fft, C, C] complex128, n_iter, status, summary)."""
    max_iterations = check_max_iterations(max_iterations)
    lib = _lib.load()
    src = accum if accum is not None else spectra
    dev = src.device
    work, nbytes = _mvar_workspace(n_groups, n_signals, n_fft, dev)
    G = torch.empty((n_groups, n_fft, n_signals, n_signals), dtype=torch.complex128, device=dev)
    n_iter = torch.empty((n_groups,), dtype=torch.int32, device=dev)
    status = torch.empty((n_groups,), dtype=torch.int32, device=dev)
    summary = (ctypes.c_int32 * 3)(0, 0, 0)
    _lib.check(lib.sc_mvar_factor_f64(_ptr(accum) if accum is not None else None,
                                      _ptr(spectra) if spectra is not None else None, n_groups, n_freq_accum, n_fft,
                                      n_signals, rec_planes(accum, planes) if accum is not None else planes, n_obs,
                                      tolerance, max_iterations, _ptr(work), nbytes, _ptr(G),
                                      _ptr(n_iter), _ptr(status), summary, _stream()), "sc_mvar_factor_f64")
    return G, n_iter, status, (summary[0], summary[1], summary[2])


def mvar_measure(G, which):
    """A directed MVAR measure / model quantity (``_lib.MVAR_*``) from the minimum-phase factor G."""
    lib = _lib.load()
    n_groups, n_fft, C, _ = G.shape
    F = n_fft // 2 + 1
    work, nbytes = _mvar_workspace(n_groups, C, n_fft, G.device)
    if which == _lib.MVAR_NOISE_COVARIANCE:
        out = torch.empty((n_groups, C, C), dtype=torch.float64, device=G.device)
    elif which in (_lib.MVAR_TRANSFER, _lib.MVAR_COEFFICIENTS):
        out = torch.empty((n_groups, F, C, C), dtype=torch.complex128, device=G.device)
    else:
        out = torch.empty((n_groups, F, C, C), dtype=torch.float64, device=G.device)
    _lib.check(lib.sc_mvar_measure_f64(_ptr(G), n_groups, n_fft, C, which, _ptr(out), _ptr(work), nbytes, _stream()),
               "sc_mvar_measure_f64")
    return out


def global_coherence(accum, n_groups, n_freq_accum, n_fft, n_signals, planes, n_obs, max_rank, ascending):
    """Leading eigenpairs of the CSM per (window, two-sided bin) (sc_global.hip)."""
    lib = _lib.load()
    dev = accum.device
    values = torch.empty((n_groups, n_fft, max_rank), dtype=torch.float64, device=dev)
    vectors = torch.empty((n_groups, n_fft, n_signals, max_rank), dtype=torch.complex128, device=dev)
    _lib.check(lib.sc_global_coherence_f64(_ptr(accum), n_groups, n_freq_accum, n_fft, n_signals,
                                           rec_planes(accum, planes), n_obs,
                                           max_rank, int(ascending), _ptr(values), _ptr(vectors), _stream()),
               "sc_global_coherence_f64")
    return values, vectors


def canonical_coherence(accum, n_signals, planes, n_obs, groups):
    """groups: list of int arrays (channel indices per group).  Returns ([n_bins, G, G] float64, n_fail)."""
    lib = _lib.load()
    dev = accum.device
    G = len(groups)
    cmax = max(len(g) for g in groups)
    stride = 16 if cmax <= 16 else (32 if cmax <= 32 else 128)      # member-table stride of the kernel that takes this size
    members = np.full((G, stride), -1, dtype=np.int32)
    for i, g in enumerate(groups):
        members[i, : min(len(g), stride)] = np.asarray(g, dtype=np.int32)[:stride]
    sizes = np.array([len(g) for g in groups], dtype=np.int32)
    members_t, sizes_t = torch.from_numpy(members).to(dev), torch.from_numpy(sizes).to(dev)
    n_bins = accum.shape[0]
    out = torch.empty((n_bins, G, G), dtype=torch.float64, device=dev)
    fail = torch.zeros((1,), dtype=torch.int32, device=dev)
    _lib.check(lib.sc_canonical_coherence_f64(_ptr(accum), n_bins, n_signals, rec_planes(accum, planes), n_obs,
                                              _ptr(members_t),
                                              _ptr(sizes_t), G, int(cmax), _ptr(out), _ptr(fail), _stream()),
               "sc_canonical_coherence_f64")
    return out, int(fail.item())


class GraphedMeasures:
    """Stage A, stage B and the epilogue of ONE fixed request, captured once in a hipGraph and replayed per time series.

    A small request -- BASELINE configs[1]: 32 channels x 100 trials x 1024 samples, 82 us of kernels in three launches -- is bound
    by the host: every launch costs the CPU 5-10 us, and the eager pass takes 0.11 ms for 0.08 ms of device work.  Captured once
    (the kernels, their arguments, the buffers they run on), the pass replays with ONE launch; the results are bit-identical to
    the eager pass (tests/test_gpu_configs.py::test_graphed_measures_replay_equals_the_eager_pass).  This is for callers that
    evaluate the same geometry over and over (a sliding analysis of a stream, a parameter scan over data sets of one shape): the
    input and the results live in buffers the object owns.

        g = engine.GraphedMeasures((T, R, C), tapers_over_fs, n_window, n_step, n_fft, "constant", "trials_tapers",
                                   [_lib.M_COHERENCY])
        out, = g(x)          # x: (T, R, C) float32 tensor (device, or host: copied in); out is overwritten by the next call

    float32 engine, complex64 spectra (the planes format starts at 256 MB of spectra, far above what a graph helps with).
    """

    def __init__(self, shape, tapers_over_fs, n_window, n_step, n_fft, detrend_type, expectation_type, measures, device=None):
        T, R, C = (int(v) for v in shape)
        dev = tapers_over_fs.device if device is None else torch.device(device)
        self.x = torch.zeros((T, R, C), dtype=torch.float32, device=dev)
        self.measures = list(measures)
        self.planes = 0
        for w in self.measures:
            self.planes |= _lib.MEASURE_PLANES[w]
        n_windows = int(np.floor(T / n_step - n_window / n_step + 1))
        h = tapers_over_fs.to(dev)
        self._ws = {}                                 # the split-bin scratch of the captured stage B: this object's own

        def run():
            sp = multitaper_spectra(self.x, h, n_window, n_step, n_fft, n_windows, detrend_type)
            accum, n_obs = accumulate(sp, expectation_type, self.planes, ws_owner=self._ws)
            return measure_multi(accum, C, self.planes, n_obs, self.measures)

        self._eager = run
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                 # twiddles, function attributes, the allocator's blocks: outside the capture
            run()
            run()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = run()

    def __call__(self, x=None):
        if x is not None:
            self.x.copy_(torch.as_tensor(x), non_blocking=True)
        self.graph.replay()
        return self.out

    def eager(self, x=None):
        """The same pass launch by launch (for comparison)."""
        if x is not None:
            self.x.copy_(torch.as_tensor(x), non_blocking=True)
        return self._eager()
