"""The thin host BASELINE.json's north_star names: ctypes + NumPy over the C ABI of ``include/sc_hip.h`` -- no torch.

``spectral_connectivity_amd.Connectivity`` sits on a PyTorch host (torch owns HBM buffers, streams and the RCCL
collectives of the multi-GPU path).  This module drives the SAME library with nothing but NumPy arrays: device and
page-locked memory, copies and the stream come from the library's own ``sc_device_alloc`` / ``sc_host_alloc`` /
``sc_memcpy_*`` / ``sc_stream_*`` entry points (sc_memory.hip), the way the reference's CuPy backend uploads with
``xp.asarray`` and downloads with ``.get()`` (reference transforms.py:405-439, connectivity.py:31-65).

    from spectral_connectivity_amd.numpy_host import NumpyHost
    host = NumpyHost()
    out = host.connectivity(time_series, sampling_frequency=1000, time_halfbandwidth_product=4,
                            n_time_samples_per_window=256, n_time_samples_per_step=128,
                            measures=("coherence_magnitude", "weighted_phase_lag_index"))

Scope: the float32 engine's hot path -- stage A (fused transform, or tapered windows + rocFFT for the lengths the fused
kernel does not take), stage B (every accumulator plane), the expectation-type measures of the reference
(connectivity.py:612-1159), ``expectation_type`` as in the reference.  Results are float64 / complex128 NumPy arrays
shaped like the reference's; plus, since round 4, stage D on float32 records: ``pairwise_spectral_granger_prediction`` (batched
2 x 2 Wilson) and ``canonical_coherence``.  Everything else (full Wilson / MVAR measures, global coherence, the float64 engine,
multi-GPU) lives on the PyTorch host.  One process uses one host: see _lib.load().
"""
import ctypes
from ctypes import byref, c_int32, c_int64, c_size_t, c_void_p

import numpy as np

from . import _lib
from ._lib import SpectraDesc

EXPECTATION_AXES = _lib.EXPECTATION_AXES
MEASURES = {
    "power": _lib.M_POWER, "coherency": _lib.M_COHERENCY, "coherence_magnitude": _lib.M_COHERENCE_MAGNITUDE,
    "coherence_phase": _lib.M_COHERENCE_PHASE, "imaginary_coherence": _lib.M_IMAGINARY_COHERENCE,
    "phase_locking_value": _lib.M_PLV, "phase_lag_index": _lib.M_PLI, "weighted_phase_lag_index": _lib.M_WPLI,
    "debiased_squared_phase_lag_index": _lib.M_DEBIASED_PLI2,
    "debiased_squared_weighted_phase_lag_index": _lib.M_DEBIASED_WPLI2, "pairwise_phase_consistency": _lib.M_PPC,
}


class DeviceBuffer:
    """``n_bytes`` of HBM.  Blocks come from the library's stream-ordered pool ONCE and are then recycled by this host: a
    dropped buffer waits in ``host._released`` until the host has synchronised its stream, and only then serves the next
    request of its size class.  (Handing blocks back to the driver's pool with hipFreeAsync and taking them out again in
    stream order -- legal, and what this class did first -- gave intermittently corrupted measure buffers under
    /opt/rocm 7.2's runtime as soon as a large block was carved up differently from call to call (planes-format spectra
    where the previous call's output lay); the same sequence on the runtime PyTorch ships ran clean.  Recycling whole blocks
    at synchronisation points does not depend on either.)"""

    def __init__(self, host, n_bytes):
        self._host, self.n_bytes = host, int(n_bytes)
        self._size_class = DeviceBuffer.size_class(self.n_bytes)
        free = host._free_blocks.get(self._size_class)
        if free:
            self.ptr = c_void_p(free.pop())
            return
        p = c_void_p()
        _lib.check(host.lib.sc_device_alloc(byref(p), self._size_class, host.stream), "sc_device_alloc")
        self.ptr = p

    @staticmethod
    def size_class(n_bytes):
        """Request rounded up to 512 bytes below 1 MB and to 1/16 of its power of two above (<= 6 % of slack)."""
        n = max(int(n_bytes), 1)
        if n <= (1 << 20):
            return -(-n // 512) * 512
        step = 1 << (n.bit_length() - 5)
        return -(-n // step) * step

    def free(self):
        if self.ptr is not None and self.ptr.value:
            self._host._released.append((self._size_class, self.ptr.value))     # reusable after the next synchronize()
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedArray(np.ndarray):
    """A NumPy array over page-locked memory of sc_host_alloc.  Page-locking is the expensive part (20 ms per 100 MB
    on the MI355X host, 12 ms to release), so the block goes back to a small pool with the last view of the array and
    the next result of that size reuses it: a steady-state download costs the copy alone (57 GB/s)."""

    _pool = {}                      # n_bytes -> [address, ...] of released blocks
    _pooled_bytes = 0
    POOL_LIMIT = 4 << 30

    @classmethod
    def empty(cls, lib, shape, dtype):
        dtype = np.dtype(dtype)
        n_bytes = max(int(np.prod(shape, dtype=np.int64)) * dtype.itemsize, 1)
        free = cls._pool.get(n_bytes)
        if free:
            address = free.pop()
            cls._pooled_bytes -= n_bytes
        else:
            p = c_void_p()
            _lib.check(lib.sc_host_alloc(byref(p), n_bytes), "sc_host_alloc")
            address = p.value
        raw = (ctypes.c_char * n_bytes).from_address(address)
        arr = np.frombuffer(raw, dtype=dtype, count=n_bytes // dtype.itemsize).reshape(shape).view(cls)
        arr._owner = _PinnedOwner(lib, address, n_bytes)
        return arr

    def __array_finalize__(self, obj):
        self._owner = getattr(obj, "_owner", None)

    @classmethod
    def trim(cls, lib):
        """Release the pooled blocks."""
        for blocks in cls._pool.values():
            for address in blocks:
                lib.sc_host_free(c_void_p(address))
        cls._pool.clear()
        cls._pooled_bytes = 0


class _PinnedOwner:
    def __init__(self, lib, address, n_bytes):
        self.lib, self.address, self.n_bytes = lib, address, n_bytes

    def __del__(self):
        try:
            if PinnedArray._pooled_bytes + self.n_bytes <= PinnedArray.POOL_LIMIT:
                PinnedArray._pool.setdefault(self.n_bytes, []).append(self.address)
                PinnedArray._pooled_bytes += self.n_bytes
            else:
                self.lib.sc_host_free(c_void_p(self.address))
        except Exception:
            pass


class NpSpectra(dict):
    """Device spectra of this host: a dict (X / P / scale: DeviceBuffer or None; F, W, R, K, C, C_alloc, N; f64; real_input;
    strides = (frequency, window, trial, taper) in elements or None for the dense [F][W][R][K][C_alloc] layout) with attribute
    access and the mark Connectivity looks for."""
    is_device_spectra = True

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    @property
    def n_fft(self):
        return self["N"]

    def free(self):
        for key in ("X", "P", "scale"):
            if self.get(key) is not None:
                self[key].free()
                self[key] = None


class NumpyHost:
    """One stream on the current device, buffers from the library, NumPy in and out."""

    def __init__(self):
        self.lib = _lib.load(torch_host=False)
        if _lib.gpu_switch() is False:
            raise RuntimeError(f"{_lib.ENABLE_GPU_ENV} selects the reference's NumPy backend, which this package does "
                               "not have: every computation runs on the HIP engine.")
        if _lib.device_count() < 1:
            raise RuntimeError("spectral_connectivity_amd: no ROCm GPU is visible. This engine has no CPU fallback; "
                               "run on an MI355X host.")
        s = c_void_p()
        _lib.check(self.lib.sc_stream_create(byref(s)), "sc_stream_create")
        self.stream = s
        self._twiddles = {}
        self._free_blocks, self._released = {}, []      # DeviceBuffer's block cache: size class -> [address], and the not-yet-safe ones

    def close(self):
        if self.stream is not None:
            self._twiddles.clear()
            self.trim()
            PinnedArray.trim(self.lib)
            self.lib.sc_stream_destroy(self.stream)
            self.stream = None

    def synchronize(self):
        _lib.check(self.lib.sc_stream_synchronize(self.stream), "sc_stream_synchronize")
        # nothing queued can touch the buffers dropped so far any more: they may serve new requests
        released, self._released = self._released, []
        for size_class, address in released:
            self._free_blocks.setdefault(size_class, []).append(address)
        if sum(k * len(v) for k, v in self._free_blocks.items()) > self.CACHE_LIMIT:
            self.trim()                                  # (many different shapes through one host: start over)

    CACHE_LIMIT = 64 << 30

    def trim(self):
        """Hand the cached device blocks back to the library's pool."""
        _lib.check(self.lib.sc_stream_synchronize(self.stream), "sc_stream_synchronize")
        for size_class, address in self._released:
            self._free_blocks.setdefault(size_class, []).append(address)
        self._released = []
        for blocks in self._free_blocks.values():
            for address in blocks:
                self.lib.sc_device_free(c_void_p(address), self.stream)
        self._free_blocks.clear()

    # ---- memory -------------------------------------------------------------------------------------------------
    def alloc(self, n_bytes):
        return DeviceBuffer(self, n_bytes)

    def upload(self, array):
        """Contiguous NumPy array -> device buffer (asynchronous when the array is page-locked)."""
        pinned = isinstance(array, PinnedArray) and array.flags.c_contiguous     # (ascontiguousarray returns a base-class array)
        a = np.ascontiguousarray(array)
        buf = self.alloc(a.nbytes)
        _lib.check(self.lib.sc_memcpy_h2d(buf.ptr, a.ctypes.data_as(c_void_p), a.nbytes, self.stream), "sc_memcpy_h2d")
        if not pinned:
            self.synchronize()             # a pageable source may be reused by the caller as soon as this returns
        return buf

    def download(self, buf, shape, dtype):
        """Device buffer -> NumPy array in page-locked memory (the copy runs at link rate)."""
        out = PinnedArray.empty(self.lib, shape, dtype)
        _lib.check(self.lib.sc_memcpy_d2h(out.ctypes.data_as(c_void_p), buf.ptr, out.nbytes, self.stream), "sc_memcpy_d2h")
        self.synchronize()
        return out

    def has_nonfinite(self, buf, n, f64=False):
        """The constructor's NaN / infinity scan (reference transforms.py:746-753) on the uploaded series."""
        flag = self.alloc(4)
        _lib.check(self.lib.sc_memset_zero(flag.ptr, 4, self.stream), "sc_memset_zero")
        fn = self.lib.sc_nonfinite_f64 if f64 else self.lib.sc_nonfinite_f32
        _lib.check(fn(buf.ptr, n, flag.ptr, self.stream), "sc_nonfinite")
        return bool(self.download(flag, (1,), np.int32)[0])

    # ---- stage A ------------------------------------------------------------------------------------------------
    def spectra(self, multitaper, planes_hint=None):
        """Stage A for a ``transforms.Multitaper`` (host geometry, tapers): dict with the device spectra
        X[F][W][R][K][C_alloc] complex64 and their sizes -- or, when the accumulator families ``planes_hint`` the caller will
        ask for take it (``_lib.planes_format_applies``), the same coefficients in the planes format of sc_fused2.hip:
        P (rows of sc_planes_row_bytes) + the per-channel scales, X = None."""
        import warnings
        m, lib = multitaper, self.lib
        if np.iscomplexobj(m.time_series):
            raise TypeError("complex-valued time series: use the PyTorch host (spectral_connectivity_amd.Multitaper), which "
                            "transforms the real and imaginary parts and assembles the two-sided spectrum")
        if m.detrend_type not in _lib.DETREND:
            raise ValueError(f"Invalid trend type '{m.detrend_type}' is not supported.\n"
                             "Valid options are 'linear'/'l', 'constant'/'c' or None.")
        ts = np.asarray(m.time_series)
        T, R, C = ts.shape
        C_alloc = C + 1 if (C % 2 and C + 1 <= _lib.PLANES_FORMAT_MAX_CHANNELS) else C       # (beyond 256 signals: planes-format requests only)
        L, step, N, W = m.n_time_samples_per_window, m.n_time_samples_per_step, m.n_fft_samples, m.n_time_windows
        tapers = np.asarray(m.tapers, dtype=np.float64)                                   # (L, K), * sqrt(fs)
        K = tapers.shape[1]
        h = self.upload(np.ascontiguousarray(tapers.T / m.sampling_frequency, dtype=np.float32))
        if ts.dtype == np.float64 and ts.size:
            # float64 series: converted on the device, the per-(trial, signal) constant taken out in float64 first
            xd = self.upload(ts)
            x = self.alloc(T * R * C_alloc * 4)
            _lib.check(lib.sc_timeseries_to_f32(xd.ptr, T, R, C, int(m.detrend_type is not None), x.ptr, C_alloc,
                                                self.stream), "sc_timeseries_to_f32")
            xd.free()
        else:
            xh = np.ascontiguousarray(ts, dtype=np.float32)
            if C_alloc != C:
                xh = np.concatenate([xh, np.zeros(xh.shape[:2] + (1,), dtype=np.float32)], axis=2)
            x = self.upload(xh)
        if getattr(m, "_finite_checked", True) is False and self.has_nonfinite(x, T * R * C_alloc):
            warnings.warn("Input time_series contains NaN or infinite values.\n"
                          "This will produce invalid spectral estimates.", UserWarning, stacklevel=3)
        F = N // 2 + 1
        detrend = _lib.DETREND[m.detrend_type]
        fused = bool(lib.sc_multitaper_fft_supported(L, N))
        if fused and N not in self._twiddles:
            tw = self.alloc(N * 8)
            _lib.check(lib.sc_fft_twiddles_f32(N, tw.ptr, self.stream), "sc_fft_twiddles_f32")
            self._twiddles[N] = tw
        if fused and _lib.planes_format_applies(L, N, C_alloc, planes_hint, spectra_bytes=F * W * R * K * C_alloc * 8):
            # planes format: one scan of the series for the channel scales, then the fused transform writes the f16 pieces
            P = self.alloc(F * W * R * K * int(lib.sc_planes_row_bytes(C_alloc)))
            work_bytes = int(lib.sc_planes_scales_work_bytes(T * R, C_alloc))
            scale, work, rng = self.alloc(2 * C_alloc * 4), self.alloc(work_bytes), self.alloc(4)
            h32 = np.asarray(tapers.T / m.sampling_frequency, dtype=np.float32)
            h_abs_sum, h_l2_min = float(np.abs(h32).sum(axis=1).max()), float(np.sqrt((h32.astype(np.float64) ** 2).sum(axis=1)).min())
            _lib.check(lib.sc_planes_scales_quality_f32(x.ptr, T, R, C_alloc, detrend, h_abs_sum, scale.ptr, work.ptr, work_bytes, rng.ptr,
                                                        self.stream), "sc_planes_scales_quality_f32")
            _lib.check(lib.sc_multitaper_fft_planes_f32(x.ptr, T, R, C_alloc, L, step, W, N, h.ptr, K, detrend,
                                                        self._twiddles[N].ptr, scale.ptr, P.ptr, self.stream),
                       "sc_multitaper_fft_planes_f32")
            # the quality check of the format (Multitaper.device_spectra of the PyTorch host does the same): one scale per channel
            # serves every window, so a channel with samples far outside its usual range keeps complex64
            ratio = float(self.download(rng, (1,), np.float32)[0]) * h_l2_min
            work.free(); rng.free()
            if ratio >= _lib.PLANES_MIN_TYPICAL:
                for b in (x, h):
                    b.free()
                return NpSpectra(X=None, P=P, scale=scale, F=F, W=W, R=R, K=K, C=C, C_alloc=C_alloc, N=N, f64=False, real_input=True,
                                 strides=None)
            P.free(); scale.free()
        X = self.alloc(F * W * R * K * C_alloc * 8)
        if fused:
            _lib.check(lib.sc_multitaper_fft_f32(x.ptr, T, R, C_alloc, L, step, W, N, h.ptr, K, detrend,
                                                 self._twiddles[N].ptr, X.ptr, self.stream), "sc_multitaper_fft_f32")
        else:
            batch = W * R * K * C_alloc
            y = self.alloc(batch * N * 4)
            _lib.check(lib.sc_taper_windows_f32(x.ptr, T, R, C_alloc, L, step, W, N, h.ptr, K, detrend, y.ptr,
                                                self.stream), "sc_taper_windows_f32")
            plan = c_void_p()
            _lib.check(lib.sc_fft_plan_create(byref(plan), N, batch), "sc_fft_plan_create")
            try:
                _lib.check(lib.sc_fft_execute(plan, y.ptr, X.ptr, self.stream), "sc_fft_execute")
                self.synchronize()
            finally:
                lib.sc_fft_plan_destroy(plan)
            y.free()
        x.free()
        h.free()
        return NpSpectra(X=X, P=None, scale=None, F=F, W=W, R=R, K=K, C=C, C_alloc=C_alloc, N=N, f64=False, real_input=True, strides=None)

    def spectra_f64(self, multitaper):
        """Stage A of the float64 engine (the reference's default dtype): float64 windows, tapers and transform, complex128
        spectra X[F][W][R][K][C] -- one fused kernel (sc_multitaper_fft_f64) for the lengths it has, sc_taper_windows_f64 +
        double-precision rocFFT otherwise (engine.multitaper_spectra_f64 of the PyTorch host)."""
        import warnings
        m, lib = multitaper, self.lib
        if np.iscomplexobj(m.time_series):
            raise TypeError("complex-valued time series: use the PyTorch host (SC_HIP_HOST=torch)")
        if m.detrend_type not in _lib.DETREND:
            raise ValueError(f"Invalid trend type '{m.detrend_type}' is not supported.\n"
                             "Valid options are 'linear'/'l', 'constant'/'c' or None.")
        ts = np.ascontiguousarray(np.asarray(m.time_series), dtype=np.float64)
        T, R, C = ts.shape
        L, step, N, W = m.n_time_samples_per_window, m.n_time_samples_per_step, m.n_fft_samples, m.n_time_windows
        tapers = np.asarray(m.tapers, dtype=np.float64)
        K = tapers.shape[1]
        x = self.upload(ts)
        h = self.upload(np.ascontiguousarray(tapers.T / m.sampling_frequency, dtype=np.float64))
        if getattr(m, "_finite_checked", True) is False:
            m._finite_checked = True
            if self.has_nonfinite(x, T * R * C, f64=True):
                warnings.warn("Input time_series contains NaN or infinite values.\n"
                              "This will produce invalid spectral estimates.", UserWarning, stacklevel=3)
        F = N // 2 + 1
        detrend = _lib.DETREND[m.detrend_type]
        X = self.alloc(F * W * R * K * C * 16)
        if bool(lib.sc_multitaper_fft_f64_supported(L, N)) and R <= 65535 and W <= 65535:
            _lib.check(lib.sc_multitaper_fft_f64(x.ptr, T, R, C, L, step, W, N, h.ptr, K, detrend, X.ptr, self.stream),
                       "sc_multitaper_fft_f64")
        else:
            batch = W * R * K * C
            y = self.alloc(batch * N * 8)
            _lib.check(lib.sc_taper_windows_f64(x.ptr, T, R, C, L, step, W, N, h.ptr, K, detrend, y.ptr, self.stream),
                       "sc_taper_windows_f64")
            plan = c_void_p()
            _lib.check(lib.sc_fft_plan_create_f64(byref(plan), N, batch), "sc_fft_plan_create_f64")
            try:
                _lib.check(lib.sc_fft_execute_f64(plan, y.ptr, X.ptr, self.stream), "sc_fft_execute_f64")
                self.synchronize()
            finally:
                lib.sc_fft_plan_destroy(plan)
            y.free()
        x.free()
        h.free()
        return NpSpectra(X=X, P=None, scale=None, F=F, W=W, R=R, K=K, C=C, C_alloc=C, N=N, f64=True, real_input=True, strides=None)

    def upload_coefficients(self, coef, f64=False):
        """Reference-layout (W, R, K, N, C) complex coefficients -> device spectra that hold all N bins as given
        (engine.upload_coefficients of the PyTorch host: same strides, the zero pad channel of an odd count in the float32 engine)."""
        coef = np.asarray(coef)
        W, R, K, N, C = coef.shape
        if f64:
            X = self.upload(np.ascontiguousarray(coef, dtype=np.complex128))
            return NpSpectra(X=X, P=None, scale=None, F=N, W=W, R=R, K=K, C=C, C_alloc=C, N=N, f64=True, real_input=False,
                             strides=(C, R * K * N * C, K * N * C, N * C))
        coef = np.ascontiguousarray(coef, dtype=np.complex64)
        if C % 2 and C + 1 <= 256:
            coef = np.concatenate([coef, np.zeros(coef.shape[:-1] + (1,), dtype=np.complex64)], axis=-1)
        Ca = coef.shape[-1]
        X = self.upload(coef)
        return NpSpectra(X=X, P=None, scale=None, F=N, W=W, R=R, K=K, C=C, C_alloc=Ca, N=N, f64=False, real_input=False,
                         strides=(Ca, R * K * N * Ca, K * N * Ca, N * Ca))

    # ---- stages B and C -----------------------------------------------------------------------------------------
    @staticmethod
    def _desc(sp, expectation_type, padded, n_freq=None):
        axes = EXPECTATION_AXES[expectation_type]
        W, R, K, Ca = sp["W"], sp["R"], sp["K"], sp["C_alloc"]
        sF, sW, sR, sK = sp.get("strides") or (W * R * K * Ca, R * K * Ca, K * Ca, Ca)
        return SpectraDesc(n_freq=sp["F"] if n_freq is None else n_freq, n_windows=W, n_trials=R, n_tapers=K,
                           n_signals=Ca if padded else sp["C"], stride_freq=sF, stride_window=sW, stride_trial=sR, stride_taper=sK,
                           reduce_window=int(0 in axes), reduce_trial=int(1 in axes), reduce_taper=int(2 in axes),
                           reserved=0)

    def accumulate(self, sp, expectation_type, planes, n_freq=None):
        """Stage B: un-normalised records [n_bins][floats_per_bin] on the device -- float32, or float64 from complex128 spectra
        (the float64 engine: sc_accumulate_f64).  ``n_freq``: accumulate the first n_freq bins only."""
        lib = self.lib
        if sp["C"] > 256 and sp.get("P") is None:
            raise ValueError(f"one launch of the complex64 / float64 stage-B kernels takes n_signals <= 256 (got {sp['C']}): Connectivity of this host "
                             "tiles more signals into channel blocks (numpy_api.Connectivity); NumpyHost.accumulate does not")
        d_real, d_pad = self._desc(sp, expectation_type, False, n_freq), self._desc(sp, expectation_type, True, n_freq)
        n_bins, fpb, n_groups, n_obs = c_int64(), c_int64(), c_int64(), c_int64()
        _lib.check(lib.sc_accum_layout(byref(d_real), planes, byref(n_bins), byref(fpb), byref(n_groups), byref(n_obs)),
                   "sc_accum_layout")
        if sp.get("f64"):
            accum = self.alloc(n_bins.value * fpb.value * 8)
            _lib.check(lib.sc_accumulate_f64(sp["X"].ptr, byref(d_real), planes, planes, accum.ptr, self.stream), "sc_accumulate_f64")
            return accum, n_bins.value, n_obs.value
        accum = self.alloc(n_bins.value * fpb.value * 4)
        if sp.get("P") is not None:
            # planes format (the families planes_format_applies admits are exactly what sc_fused2.hip accumulates)
            if not lib.sc_fused2_supported(byref(d_pad), planes):
                raise _lib.HipEngineError("planes-format spectra: this expectation type / plane set needs complex64 spectra "
                                          "(call spectra() without planes_hint)")
            ws_bytes = int(lib.sc_fused_workspace_bytes(byref(d_pad), planes))
            ws = self.alloc(ws_bytes) if ws_bytes else None
            _lib.check(lib.sc_fused2_csm_absim_f32(sp["P"].ptr, byref(d_pad), sp["scale"].ptr, planes, accum.ptr,
                                                   ws.ptr if ws else None, ws_bytes, self.stream), "sc_fused2_csm_absim_f32")
            if ws:
                ws.free()
            return accum, n_bins.value, n_obs.value
        one_pass = int(lib.sc_fused_planes_covered(byref(d_pad), planes)) if lib.sc_fused_supported(sp["C_alloc"]) else 0
        X, st = sp["X"].ptr, self.stream
        if one_pass:
            ws_bytes = int(lib.sc_fused_workspace_bytes(byref(d_pad), planes))
            ws = self.alloc(ws_bytes) if ws_bytes else None
            ws_ptr = ws.ptr if ws else None
            if one_pass & _lib.PLANE_CSM:
                _lib.check(lib.sc_fused_csm_absim_ws_f32(X, byref(d_pad), planes, accum.ptr, ws_ptr, ws_bytes, st),
                           "sc_fused_csm_absim_ws_f32")
            if one_pass & _lib.PLANE_SIGN_IM:
                _lib.check(lib.sc_fused_sign_ws_f32(X, byref(d_pad), planes, accum.ptr, ws_ptr, ws_bytes, st),
                           "sc_fused_sign_ws_f32")
            if one_pass & _lib.PLANE_UNIT:
                _lib.check(lib.sc_fused_unit_ws_f32(X, byref(d_pad), planes, accum.ptr, ws_ptr, ws_bytes, None, 0, st),
                           "sc_fused_unit_ws_f32")
            if ws:
                ws.free()
        elif planes & _lib.PLANE_CSM:
            _lib.check(lib.sc_csm_accumulate_f32(X, byref(d_real), planes, accum.ptr, st), "sc_csm_accumulate_f32")
            one_pass = _lib.PLANE_CSM
        rest = planes & ~one_pass
        if rest:
            _lib.check(lib.sc_nonlinear_accumulate_f32(X, byref(d_real), planes, rest, accum.ptr, st),
                       "sc_nonlinear_accumulate_f32")
        return accum, n_bins.value, n_obs.value

    # ---- stage D through this host: pairwise spectral Granger, canonical coherence -------------------------------------
    def _csm_records(self, time_series, expectation_type, multitaper_kwargs):
        from .transforms import Multitaper
        if expectation_type not in EXPECTATION_AXES:
            raise ValueError(f"Invalid expectation_type '{expectation_type}'. Must be one of: "
                             + ", ".join(f"'{k}'" for k in EXPECTATION_AXES))
        m = Multitaper(time_series, **multitaper_kwargs)
        if np.asarray(m.time_series).shape[2] > 256:      # (before anything is allocated on the device)
            raise ValueError(f"n_signals <= 256 through NumpyHost's functional interface (got {np.asarray(m.time_series).shape[2]}): "
                             "numpy_api.Connectivity (SC_HIP_HOST=numpy) tiles more signals into channel blocks")
        planes = _lib.PLANE_CSM
        # (complex64 spectra, like Connectivity._csm_records of the PyTorch host: these consumers read every bin of the CSM once, at
        #  window lengths and channel counts where the planes format buys nothing)
        sp = self.spectra(m, planes_hint=None)
        accum, n_bins, n_obs = self.accumulate(sp, expectation_type, planes)
        for key in ("X", "P", "scale"):
            if sp.get(key) is not None:
                sp[key].free()
        axes = EXPECTATION_AXES[expectation_type]
        kept = tuple(n for i, n in enumerate((sp["W"], sp["R"], sp["K"])) if i not in axes)
        return m, sp, accum, n_bins, n_obs, kept

    # ---- SURVEY 8(f) through this host: the full Wilson factor + the directed MVAR measures, global coherence ------------------
    MVAR_MEASURES = {"directed_transfer_function": _lib.MVAR_DTF, "directed_coherence": _lib.MVAR_DC,
                     "partial_directed_coherence": _lib.MVAR_PDC, "generalized_partial_directed_coherence": _lib.MVAR_GPDC,
                     "direct_directed_transfer_function": _lib.MVAR_DDTF}

    def mvar_measures(self, time_series, measures=("directed_transfer_function",), expectation_type="trials_tapers", tolerance=1e-8,
                      max_iterations=60, **multitaper_kwargs):
        """NumPy time series -> {name: array} for the reference's directed measures of the full multivariate model
        (``Connectivity.directed_transfer_function`` ... ``direct_directed_transfer_function``, connectivity.py:1237-1426; out[...,
        i, j] = j -> i on the non-negative bins): cross-spectral records on the device, ONE batched C x C Wilson factorisation
        (sc_mvar_factor_f64, connectivity.py:567-589), one small kernel and one download per measure."""
        unknown = [name for name in measures if name not in self.MVAR_MEASURES]
        if unknown:
            raise ValueError(f"unknown MVAR measures {unknown}; available: {sorted(self.MVAR_MEASURES)}")
        if not 1 <= int(max_iterations) <= 1024:
            raise ValueError("max_iterations must be in 1 ... 1024")
        m, sp, accum, n_bins, n_obs, kept = self._csm_records(time_series, expectation_type, multitaper_kwargs)
        lib, C, F, N = self.lib, sp["C"], sp["F"], sp["N"]
        if C > lib.sc_mvar_max_signals():
            accum.free()
            raise ValueError(f"the full Wilson factorisation supports n_signals <= {lib.sc_mvar_max_signals()} (got {C})")
        n_groups = n_bins // F
        nbytes = ctypes.c_size_t()
        _lib.check(lib.sc_mvar_workspace_bytes(n_groups, C, N, byref(nbytes)), "sc_mvar_workspace_bytes")
        work = self.alloc(nbytes.value)
        G = self.alloc(n_groups * N * C * C * 16)
        n_iter, status = self.alloc(n_groups * 4), self.alloc(n_groups * 4)
        summary = (ctypes.c_int32 * 3)(0, 0, 0)
        _lib.check(lib.sc_mvar_factor_f64(accum.ptr, None, n_groups, F, N, C, _lib.PLANE_CSM, n_obs, tolerance, int(max_iterations),
                                          work.ptr, nbytes.value, G.ptr, n_iter.ptr, status.ptr, summary, self.stream), "sc_mvar_factor_f64")
        self.last_wilson = dict(iterations=int(summary[0]), not_converged=int(summary[1]), cholesky_fallbacks=int(summary[2]))
        out = {}
        for name in measures:
            dev = self.alloc(n_groups * F * C * C * 8)
            _lib.check(lib.sc_mvar_measure_f64(G.ptr, n_groups, N, C, self.MVAR_MEASURES[name], dev.ptr, work.ptr, nbytes.value, self.stream),
                       "sc_mvar_measure_f64")
            out[name] = np.array(self.download(dev, kept + (F, C, C), np.float64))
            dev.free()
        for b in (work, G, n_iter, status, accum):
            b.free()
        return out

    def global_coherence(self, time_series, max_rank=1, **multitaper_kwargs):
        """NumPy time series -> (values (n_time_windows, n_fft_samples, max_rank), vectors (n_time_windows, n_fft_samples, n_signals,
        max_rank)) like the reference's ``Connectivity.global_coherence`` (connectivity.py:822-895): the leading eigenpairs of the
        cross-spectral matrix of every (window, two-sided bin) (sc_global_coherence_f64); always over trials and tapers."""
        m, sp, accum, n_bins, n_obs, kept = self._csm_records(time_series, "trials_tapers", multitaper_kwargs)
        lib, C, F, N, W = self.lib, sp["C"], sp["F"], sp["N"], sp["W"]
        max_rank = int(max_rank)
        if not 1 <= max_rank <= min(C, sp["R"] * sp["K"]):
            accum.free()
            raise ValueError(f"max_rank must be between 1 and min(n_signals, n_trials * n_tapers) = {min(C, sp['R'] * sp['K'])}")
        if C > lib.sc_global_coherence_max_signals():
            accum.free()
            raise ValueError(f"global_coherence supports n_signals <= {lib.sc_global_coherence_max_signals()}")
        values, vectors = self.alloc(W * N * max_rank * 8), self.alloc(W * N * C * max_rank * 16)
        _lib.check(lib.sc_global_coherence_f64(accum.ptr, W, F, N, C, _lib.PLANE_CSM, n_obs, max_rank, int(max_rank < C - 1), values.ptr,
                                               vectors.ptr, self.stream), "sc_global_coherence_f64")
        res = (np.array(self.download(values, (W, N, max_rank), np.float64)),
               np.array(self.download(vectors, (W, N, C, max_rank), np.complex128)))
        for b in (values, vectors, accum):
            b.free()
        return res

    def pairwise_spectral_granger_prediction(self, time_series, pairs=None, expectation_type="trials_tapers", tolerance=1e-8,
                                             max_iterations=60, **multitaper_kwargs):
        """NumPy time series -> the reference's ``Connectivity.pairwise_spectral_granger_prediction()`` (connectivity.py:
        1161-1213; out[..., i, j] = j -> i, NaN elsewhere) for all channel pairs or the listed ``pairs``: cross-spectral records
        on the device, batched 2 x 2 Wilson factorisations (sc_granger_pairwise_f64), one download."""
        m, sp, accum, n_bins, n_obs, kept = self._csm_records(time_series, expectation_type, multitaper_kwargs)
        lib, C, F, N = self.lib, sp["C"], sp["F"], sp["N"]
        if pairs is None:
            pairs = [(i, j) for i in range(C) for j in range(i + 1, C)]
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        if ((pairs < 0) | (pairs >= C)).any():
            raise IndexError("pair index outside the signals")
        pairs = pairs[pairs[:, 0] != pairs[:, 1]]
        if not 1 <= int(max_iterations) <= 1024:
            raise ValueError("max_iterations must be in 1 ... 1024")
        n_groups = n_bins // F
        out = self.alloc(n_groups * F * C * C * 8)
        result_shape = kept + (F, C, C)
        if len(pairs) == 0:
            accum.free(); out.free()
            return np.full(result_shape, np.nan)
        per_pair = n_groups * N * 160                                   # workspace bytes per problem (sc_granger_workspace_bytes)
        chunk = int(max(1, min(len(pairs), (8 << 30) // per_pair)))
        nbytes = ctypes.c_size_t()
        _lib.check(lib.sc_granger_workspace_bytes(n_groups, chunk, N, byref(nbytes)), "sc_granger_workspace_bytes")
        work = self.alloc(nbytes.value)
        not_converged = fallbacks = 0
        for p0 in range(0, len(pairs), chunk):
            n = min(chunk, len(pairs) - p0)
            d_pairs = self.upload(pairs[p0:p0 + n])
            it_c, st_c = self.alloc(n_groups * n * 4), self.alloc(n_groups * n * 4)
            summary = (ctypes.c_int32 * 3)(0, 0, 0)
            _lib.check(lib.sc_granger_pairwise_f64(accum.ptr, n_groups, F, N, C, _lib.PLANE_CSM, n_obs, d_pairs.ptr, n, tolerance,
                                                   int(max_iterations), work.ptr, nbytes.value, _lib.GRANGER_KEEP_OUTPUT if p0 else 0,
                                                   out.ptr, it_c.ptr, st_c.ptr, summary, self.stream), "sc_granger_pairwise_f64")
            not_converged += summary[1]
            fallbacks += summary[2]
            for b in (d_pairs, it_c, st_c):
                b.free()
        res = np.array(self.download(out, result_shape, np.float64))
        for b in (work, out, accum):
            b.free()
        self.last_wilson = dict(not_converged=int(not_converged), cholesky_fallbacks=int(fallbacks))
        return res

    def canonical_coherence(self, time_series, group_labels, **multitaper_kwargs):
        """NumPy time series -> (array (n_time_windows, n_frequencies, n_groups, n_groups), sorted labels) like the reference's
        ``Connectivity.canonical_coherence(group_labels)`` (connectivity.py:745-820, 1953-2032; always over trials and tapers)."""
        m, sp, accum, n_bins, n_obs, kept = self._csm_records(time_series, "trials_tapers", multitaper_kwargs)
        lib, C, F = self.lib, sp["C"], sp["F"]
        group_labels = np.asarray(group_labels)
        if group_labels.shape != (C,):
            raise ValueError(f"group_labels needs one label per signal ({C}), got shape {group_labels.shape}")
        labels = np.unique(group_labels)
        groups = [np.flatnonzero(group_labels == lab) for lab in labels]
        n_g = len(groups)
        res = np.ones((n_bins, n_g, n_g))
        res[:, np.arange(n_g), np.arange(n_g)] = np.nan
        # (a group with at least as many channels as observations spans the observation space: coherence 1 with every other group)
        small = [k for k, g in enumerate(groups) if len(g) < n_obs]
        max_group = int(lib.sc_canonical_max_group())
        if any(len(groups[k]) > max_group for k in small):
            raise ValueError(f"canonical_coherence: the whitening kernel takes up to {max_group} channels per group")
        if len(small) >= 2:
            cmax = max(len(groups[k]) for k in small)
            stride = 16 if cmax <= 16 else (32 if cmax <= 32 else 128)
            members = np.full((len(small), stride), -1, dtype=np.int32)
            for i, k in enumerate(small):
                members[i, :len(groups[k])] = groups[k]
            sizes = np.array([len(groups[k]) for k in small], dtype=np.int32)
            d_members, d_sizes = self.upload(members), self.upload(sizes)
            out, fail = self.alloc(n_bins * len(small) * len(small) * 8), self.alloc(4)
            _lib.check(lib.sc_memset_zero(fail.ptr, 4, self.stream), "sc_memset_zero")
            _lib.check(lib.sc_canonical_coherence_f64(accum.ptr, n_bins, C, _lib.PLANE_CSM, n_obs, d_members.ptr, d_sizes.ptr, len(small),
                                                      int(cmax), out.ptr, fail.ptr, self.stream), "sc_canonical_coherence_f64")
            sub = np.array(self.download(out, (n_bins, len(small), len(small)), np.float64))
            self.last_canonical_failures = int(self.download(fail, (1,), np.int32)[0])
            res[np.ix_(np.arange(n_bins), small, small)] = sub
            for b in (d_members, d_sizes, out, fail):
                b.free()
        accum.free()
        return res.reshape(sp["W"], F, n_g, n_g), labels

    def _planes_expectation(self, m, expectation_type, planes):
        """Would sc_fused2.hip take this request?  (asked BEFORE stage A picks the device format of the spectra)"""
        ts = np.asarray(m.time_series)
        C = ts.shape[2]
        C_alloc = C + 1 if (C % 2 and C + 1 <= 256) else C
        geom = dict(F=m.n_fft_samples // 2 + 1, W=m.n_time_windows, R=ts.shape[1], K=np.asarray(m.tapers).shape[1], C=C, C_alloc=C_alloc)
        return bool(self.lib.sc_fused2_supported(byref(self._desc(geom, expectation_type, True)), planes))

    def connectivity(self, time_series, measures=("coherence_magnitude",), expectation_type="trials_tapers", **multitaper_kwargs):
        """NumPy time series (n_time, n_trials, n_signals) -> {measure name: NumPy array} shaped like the reference's
        ``Connectivity.<measure>()`` results (non-negative frequencies)."""
        from .transforms import Multitaper
        if expectation_type not in EXPECTATION_AXES:
            raise ValueError(f"Invalid expectation_type '{expectation_type}'. Must be one of: "
                             + ", ".join(f"'{k}'" for k in EXPECTATION_AXES))
        unknown = [name for name in measures if name not in MEASURES]
        if unknown:
            raise ValueError(f"unknown measures {unknown}; available: {sorted(MEASURES)}")
        m = Multitaper(time_series, **multitaper_kwargs)
        if np.asarray(m.time_series).shape[2] > 256:      # (before anything is allocated on the device)
            raise ValueError(f"n_signals <= 256 through NumpyHost's functional interface (got {np.asarray(m.time_series).shape[2]}): "
                             "numpy_api.Connectivity (SC_HIP_HOST=numpy) tiles more signals into channel blocks")
        planes = 0
        for name in measures:
            planes |= _lib.MEASURE_PLANES[MEASURES[name]]
        # (the planes format holds observations as ONE run of rows: the expectation types that reduce every stored axis but
        #  the frequency, or a contiguous tail of them -- sc_fused2_supported decides; anything else takes complex64)
        sp = self.spectra(m, planes_hint=planes if self._planes_expectation(m, expectation_type, planes) else None)
        accum, n_bins, n_obs = self.accumulate(sp, expectation_type, planes)
        for key in ("X", "P", "scale"):
            if sp.get(key) is not None:
                sp[key].free()
        C, F = sp["C"], sp["F"]
        axes = EXPECTATION_AXES[expectation_type]
        kept = tuple(n for i, n in enumerate((sp["W"], sp["R"], sp["K"])) if i not in axes)
        out = {}
        for name in measures:
            which = MEASURES[name]
            tail = (C,) if which == _lib.M_POWER else (C, C)
            dtype = np.complex128 if which in _lib.COMPLEX_MEASURES else np.float64
            dev = self.alloc(n_bins * int(np.prod(tail)) * np.dtype(dtype).itemsize)
            _lib.check(self.lib.sc_measure_f64(accum.ptr, n_bins, C, planes, n_obs, which, dev.ptr, self.stream),
                       "sc_measure_f64")
            out[name] = self.download(dev, kept + (F,) + tail, dtype)
            dev.free()
        accum.free()
        out["frequencies"] = np.asarray(m.frequencies)[:F].copy()
        if F and out["frequencies"][-1] < 0:
            out["frequencies"][-1] = abs(out["frequencies"][-1])
        out["time"] = np.asarray(m.time)
        return out
