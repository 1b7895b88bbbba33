"""``Multitaper`` / ``Connectivity`` on the torch-free host: the reference's own dependencies (NumPy + SciPy,
``pyproject.toml:42-47``) and ``libsc_hip.so`` -- nothing else.

``spectral_connectivity_amd.Connectivity`` sits on one of two hosts of the same C ABI (``include/sc_hip.h``):

* the PyTorch host (``engine.py``): torch owns the HBM buffers, the stream and the RCCL collectives of the multi-GPU path;
* this one (``numpy_host.NumpyHost``: ctypes + NumPy over the library's own allocator, copies and stream).

``SC_HIP_HOST=numpy|torch`` selects (default: torch when it can be imported, NumPy otherwise); with ``numpy`` the package's
``Connectivity`` is the class below and ``Multitaper.fft()`` / ``Multitaper.device_spectra()`` run here -- ``torch`` is never
imported.  Same constructor, properties, methods, shapes, dtypes, warnings and errors as the PyTorch-host class (it IS that class:
only the methods that touch the device are replaced), both engines (``dtype=complex64`` -> float32 engine, planes format
included; ``complex128``, the default -> float64 engine), every expectation-type measure, pairwise / subset Granger, the full
Wilson factor and the directed MVAR measures, canonical and global coherence, the band statistics, more than 256 signals (channel
blocks of 128, tiled on the host).  Not here: complex-valued time series, multi-GPU (``parallel.ShardedConnectivity`` needs
``torch.distributed``), hipGraph replay (``engine.GraphedMeasures``).
"""
import ctypes
import warnings
from ctypes import byref
from logging import getLogger

import numpy as np

from . import _lib
from .connectivity import Connectivity as _TorchHostConnectivity
from .connectivity import _PendingSpectra

logger = getLogger(__name__)
BLOCK_SIGNALS = 128        # channel block of the > 256-signal tiling (engine.BLOCK_SIGNALS)
_host = None


def host():
    """The process-wide NumpyHost (one HIP runtime per process: see _lib.load())."""
    global _host
    if _host is None:
        from .numpy_host import NumpyHost
        _host = NumpyHost()
    return _host


class _Record:
    """Accumulator records on the device: [n_bins][floats_per_bin] float32 or float64."""

    def __init__(self, buf, n_bins, fpb, f64):
        self.buf, self.n_bins, self.fpb, self.f64 = buf, int(n_bins), int(fpb), bool(f64)

    @property
    def shape(self):
        return (self.n_bins, self.fpb)

    def planes(self, planes):
        return (planes | _lib.RECORD_F64) if self.f64 else (planes & ~_lib.RECORD_F64)

    def __del__(self):
        try:
            self.buf.free()
        except Exception:
            pass


def multitaper_spectra(m, precision, planes_hint=None):
    """Multitaper.device_spectra of this host."""
    h = host()
    if np.iscomplexobj(m.time_series):
        raise TypeError("complex-valued time series need the PyTorch host (SC_HIP_HOST=torch): it transforms the real and imaginary "
                        "parts with the real-input kernels and assembles the two-sided spectrum on the device")
    ts = np.asarray(m.time_series)
    if ts.shape[2] > 256:
        # more than 256 signals: planes-format spectra of the whole array where the format applies (round 6: sc_fused2.hip plans its
        # launches over any number of 32-channel blocks) -- otherwise no device spectra of the whole array, the record is tiled
        C = ts.shape[2]
        C_alloc = C + (C & 1)
        F, W, R, K = m.n_fft_samples // 2 + 1, int(m.n_time_windows), ts.shape[1], int(m.n_tapers)
        if (precision != "float64" and C_alloc <= _lib.PLANES_FORMAT_MAX_CHANNELS
                and _lib.planes_format_applies(m.n_time_samples_per_window, m.n_fft_samples, C_alloc, planes_hint,
                                               spectra_bytes=F * W * R * K * C_alloc * 8)):
            sp = h.spectra(m, planes_hint=planes_hint)
            if sp.get("P") is not None:
                sp["wide_source"] = (m, precision)       # (a family outside the format later: back to the tiling, see _accumulators)
                return sp
            sp.free()                                    # (the format's quality check sent the transform to complex64)
        return _WideSeries(m, precision)
    return h.spectra_f64(m) if precision == "float64" else h.spectra(m, planes_hint=planes_hint)


class _WideSeries:
    """More than 256 signals: no device spectra of the whole array (one launch of the stage-B kernels stages <= 256 channels) --
    the record is tiled from the spectra of channel-block pairs (Connectivity._accumulate_wide)."""
    is_device_spectra = True

    def __init__(self, multitaper, precision):
        self.multitaper, self.precision = multitaper, precision
        ts = np.asarray(multitaper.time_series)
        self.W, self.R, self.K = int(multitaper.n_time_windows), int(ts.shape[1]), int(multitaper.n_tapers)
        self.n_fft, self.C = int(multitaper.n_fft_samples), int(ts.shape[2])
        self.N, self.F = self.n_fft, self.n_fft // 2 + 1
        self.real_input, self.f64, self.P = True, precision == "float64", None


def fft(m):
    """Multitaper.fft() of this host: (n_time_windows, n_trials, n_tapers, n_fft_samples, n_signals) complex128, two-sided."""
    from . import options
    precision = options.engine_precision(None)
    ts = np.asarray(m.time_series)
    C = ts.shape[2]
    cols = [np.arange(c0, min(c0 + 256, C)) for c0 in range(0, C, 256)]
    parts = []
    for cc in cols:
        sub = m if len(cols) == 1 else _channel_subset_multitaper(m, cc)
        sp = host().spectra_f64(sub) if precision == "float64" else host().spectra(sub)
        dt = np.complex128 if sp["f64"] else np.complex64
        one = host().download(sp["X"], (sp["F"], sp["W"], sp["R"], sp["K"], sp["C_alloc"]), dt)[..., :sp["C"]]
        parts.append(np.array(one, dtype=np.complex128))
        sp.free()
    one = np.moveaxis(parts[0] if len(parts) == 1 else np.concatenate(parts, axis=-1), 0, 3)       # (W, R, K, F, C)
    N, F = m.n_fft_samples, one.shape[3]
    out = np.empty(one.shape[:3] + (N, one.shape[-1]), dtype=np.complex128)
    out[..., :F, :] = one
    if N > F:
        out[..., F:, :] = np.conj(one[..., N - F:0:-1, :])
    return out


def _channel_subset_multitaper(m, cols):
    """The same transform for the channels ``cols`` of the series (stage A is per channel: nothing else changes): a shallow copy of
    the object -- every resolved parameter, the tapers -- with the series replaced."""
    import copy
    _ = m.tapers                                           # (resolved once, shared by the copies)
    sub = copy.copy(m)
    sub.time_series = np.ascontiguousarray(np.asarray(m.time_series)[:, :, cols])
    sub._device_spectra, sub._deferred_checks = None, None
    sub._finite_checked = True                             # (the constructor's scan was the whole series')
    return sub


class Connectivity(_TorchHostConnectivity):
    """``spectral_connectivity_amd.Connectivity`` on the torch-free host (module docstring)."""

    @classmethod
    def from_multitaper(cls, multitaper_instance, expectation_type="trials_tapers", blocks=None, dtype=np.complex128):
        """Reference connectivity.py:366-400; the transform runs at the first request (its accumulator families decide the device
        format of the float32 engine's spectra)."""
        from . import options
        precision = options.engine_precision(dtype)
        multitaper_instance.check_device_path()
        if np.iscomplexobj(multitaper_instance.time_series):
            raise TypeError("complex-valued time series need the PyTorch host (SC_HIP_HOST=torch)")
        obj = cls(_PendingSpectra(multitaper_instance, precision), expectation_type=expectation_type,
                  time=multitaper_instance.time, frequencies=multitaper_instance.frequencies, blocks=blocks, dtype=dtype)
        obj._multitaper = multitaper_instance
        return obj

    # ---- device plumbing -----------------------------------------------------------------------------------------------------
    def _device(self, planes_hint=None, defer_checks=False):
        if self._spectra is None and self._pending is not None:
            if planes_hint is not None and not (planes_hint in _lib.PLANES_FORMAT_FAMILIES and self._planes_request_ok(planes_hint)):
                planes_hint = None
            self._spectra = multitaper_spectra(self._pending.multitaper, self._pending.precision, planes_hint)
            self._pending = None
        if self._spectra is None:
            _lib.require_gpu()
            if self._host_coefficients.shape[-1] > 256:
                raise ValueError("uploaded coefficients of more than 256 signals: build the object with Connectivity.from_multitaper "
                                 "(the channel blocks are transformed from the series) or use the PyTorch host")
            self._spectra = host().upload_coefficients(self._host_coefficients, f64=self._precision == "float64")
        return self._spectra

    def _settle(self):
        return True

    def _accumulate(self, sp, expectation_type, planes, n_freq):
        if isinstance(sp, _WideSeries):
            return self._accumulate_wide(sp, expectation_type, planes, n_freq)
        buf, n_bins, n_obs = host().accumulate(sp, expectation_type, planes, n_freq=n_freq)
        elem = 8 if sp["f64"] else 4
        return _Record(buf, n_bins, buf.n_bytes // (n_bins * elem) if n_bins else 0, sp["f64"]), n_obs

    def _accumulate_wide(self, wide, expectation_type, planes, n_freq):
        """engine._accumulate_blocked on this host: channel blocks of 128; every pair of blocks is a request of its own (<= 256
        signals: the ordinary kernels, spectra of just those channels from the series), its 16 x 16 record tiles are placed into the
        full record on the HOST, and the full record goes back to the device once for whatever consumes it."""
        C, m = wide.C, wide.multitaper
        n_blk = -(-C // BLOCK_SIGNALS)
        NB = -(-C // 16)
        n_tiles = NB * (NB + 1) // 2
        per = BLOCK_SIGNALS // 16
        full, n_obs_out, n_planes = None, None, None
        dt = np.float64 if wide.f64 else np.float32

        def tile(bi, bj, nb):
            return bi * nb - bi * (bi - 1) // 2 + (bj - bi)

        for a in range(n_blk - 1):
            for b in range(a + 1, n_blk):
                cols = np.concatenate([np.arange(a * BLOCK_SIGNALS, (a + 1) * BLOCK_SIGNALS),
                                       np.arange(b * BLOCK_SIGNALS, min((b + 1) * BLOCK_SIGNALS, C))])
                sub_m = _channel_subset_multitaper(m, cols)
                sp = host().spectra_f64(sub_m) if wide.f64 else host().spectra(sub_m)
                buf, n_bins, n_obs = host().accumulate(sp, expectation_type, planes, n_freq=n_freq)
                sp.free()
                nb_s = -(-len(cols) // 16)
                nt_s = nb_s * (nb_s + 1) // 2
                rec = np.array(host().download(buf, (n_bins, buf.n_bytes // (n_bins * dt().itemsize)), dt))
                buf.free()
                n_planes = rec.shape[1] // (nt_s * 256)
                rec = rec.reshape(n_bins, n_planes, nt_s, 256)
                if full is None:
                    full, n_obs_out = np.zeros((n_bins, n_planes, n_tiles, 256), dtype=dt), n_obs
                src, dst = [], []
                for ti in range(nb_s):
                    for tj in range(ti, nb_s):
                        in_a_i, in_a_j = ti < per, tj < per
                        if in_a_i and in_a_j:
                            keep = b == a + 1
                        elif not in_a_i and not in_a_j:
                            keep = a == n_blk - 2 and b == n_blk - 1
                        else:
                            keep = True
                        if keep:
                            gi = a * per + ti if in_a_i else b * per + (ti - per)
                            gj = a * per + tj if in_a_j else b * per + (tj - per)
                            src.append(tile(ti, tj, nb_s))
                            dst.append(tile(gi, gj, NB))
                full[:, :, dst] = rec[:, :, src]
        n_bins = full.shape[0]
        full = full.reshape(n_bins, -1)
        return _Record(host().upload(full), n_bins, full.shape[1], wide.f64), n_obs_out

    def _accumulators(self, planes, defer_checks=False):
        for have, rec in self._accum_cache.items():
            if isinstance(have, int) and have & planes == planes:
                return have, rec
        from . import options
        sp = self._device(planes_hint=planes)
        if (getattr(sp, "P", None) is not None and planes == _lib.PLANE_CSM and options.anticipate_phase_lag
                and self._planes_request_ok(_lib.PLANE_CSM | _lib.PLANE_ABS_IM)):
            planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM            # (Connectivity._accumulators: coherence then wPLI is one pass)
        elif getattr(sp, "f64", False) and planes == _lib.PLANE_CSM and options.anticipate_phase_lag and not isinstance(sp, _WideSeries):
            # float64 engine on this host: a later phase-lag request cannot copy the families a record already holds (the PyTorch
            # host does, with a strided device copy) and would accumulate everything again -- the |Im s| plane rides along instead
            planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
        if getattr(sp, "P", None) is not None and not host().lib.sc_fused2_supported(
                byref(host()._desc(sp, self.expectation_type, True, self._n_freq)), planes):
            # spectra held as f16 pieces, and a family their kernels do not take (PLV after coherence, ...): decoded once
            self._spectra = sp = self._decode_planes(sp)
        rec, n_obs = self._accumulate(sp, self.expectation_type, planes, self._n_freq)
        for old in [h for h in self._accum_cache if isinstance(h, int) and h & planes == h]:
            del self._accum_cache[old]
        self._accum_cache[planes] = (rec, n_obs)
        return planes, (rec, n_obs)

    def _decode_planes(self, sp):
        """complex64 spectra from the planes format (sc_spectra_from_planes_f32: lossless up to its 22 bits).  More than 256 signals:
        the complex64 kernels do not take them in one piece -- back to the series and the channel-block tiling."""
        h = host()
        if sp["C"] > 256:
            m, precision = sp["wide_source"]
            sp.free()
            return _WideSeries(m, precision)
        X = h.alloc(sp["F"] * sp["W"] * sp["R"] * sp["K"] * sp["C_alloc"] * 8)
        d = h._desc(sp, "trials_tapers", True)
        _lib.check(h.lib.sc_spectra_from_planes_f32(sp["P"].ptr, byref(d), sp["scale"].ptr, X.ptr, h.stream), "sc_spectra_from_planes_f32")
        out = type(sp)(sp)
        out.update(X=X, P=None, scale=None)
        sp.free()
        return out

    def _csm_records(self, tag, expectation_type=None, two_sided=True):
        sp = self._device()
        if getattr(sp, "P", None) is not None:
            self._spectra = sp = self._decode_planes(sp)
        N = self._shape5[3]
        n_freq = (sp.F if sp.real_input else N) if two_sided else self._n_freq
        key = (tag, n_freq)
        if key not in self._accum_cache:
            self._accum_cache[key] = self._accumulate(sp, expectation_type or self.expectation_type, _lib.PLANE_CSM, n_freq)
        rec, n_obs = self._accum_cache[key]
        return rec, n_obs, n_freq

    @property
    def _shape5(self):
        if self._host_coefficients is not None:
            return self._host_coefficients.shape
        if self._spectra is None and self._pending is not None:
            return self._pending.shape5
        s = self._spectra
        return (s.W, s.R, s.K, s.n_fft, s.C)

    # ---- stage C --------------------------------------------------------------------------------------------------------------
    def _measure(self, which):
        have, (rec, n_obs) = self._accumulators(_lib.MEASURE_PLANES[which])
        C = self._shape5[4]
        wide = self._wide_output(which)
        h = host()
        if which == _lib.M_POWER:
            shape, dt = (rec.n_bins, C), (np.float64 if wide else np.float32)
        elif which in _lib.COMPLEX_MEASURES:
            shape, dt = (rec.n_bins, C, C), (np.complex128 if wide else np.complex64)
        else:
            shape, dt = (rec.n_bins, C, C), (np.float64 if wide else np.float32)
        out = h.alloc(int(np.prod(shape)) * np.dtype(dt).itemsize)
        fn = h.lib.sc_measure_f64 if wide else h.lib.sc_measure_f32
        _lib.check(fn(rec.buf.ptr, rec.n_bins, C, rec.planes(have), self._n_observations_total(n_obs), which, out.ptr, h.stream),
                   "sc_measure")
        res = h.download(out, shape, dt)        # (the page-locked array itself: its owner recycles the block with the last view)
        out.free()
        tail = (C,) if which == _lib.M_POWER else (C, C)
        return res.reshape(self._kept_shape() + (self._n_freq,) + tail)

    # ---- stage D --------------------------------------------------------------------------------------------------------------
    def _granger(self, pairs):
        N, C = self._shape5[3], self._shape5[4]
        rec, n_obs, n_freq = self._csm_records("granger")
        h, lib = host(), host().lib
        n_groups = rec.n_bins // n_freq
        F = N // 2 + 1
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        out = h.alloc(n_groups * F * C * C * 8)
        if len(pairs) == 0:
            out.free()
            return np.full(self._kept_shape() + (F, C, C), np.nan)
        per_pair = n_groups * N * 160
        chunk = int(max(1, min(len(pairs), (8 << 30) // per_pair)))
        nbytes = ctypes.c_size_t()
        _lib.check(lib.sc_granger_workspace_bytes(n_groups, chunk, N, byref(nbytes)), "sc_granger_workspace_bytes")
        work = h.alloc(nbytes.value)
        iters = not_conv = fallback = 0
        n_iter_all, status_all = [], []
        for p0 in range(0, len(pairs), chunk):
            n = min(chunk, len(pairs) - p0)
            d_pairs = h.upload(pairs[p0:p0 + n])
            it_c, st_c = h.alloc(n_groups * n * 4), h.alloc(n_groups * n * 4)
            summary = (ctypes.c_int32 * 3)(0, 0, 0)
            _lib.check(lib.sc_granger_pairwise_f64(rec.buf.ptr, n_groups, n_freq, N, C, rec.planes(_lib.PLANE_CSM),
                                                   self._n_observations_total(n_obs), d_pairs.ptr, n, 1e-8, 60, work.ptr, nbytes.value,
                                                   _lib.GRANGER_KEEP_OUTPUT if p0 else 0, out.ptr, it_c.ptr, st_c.ptr, summary, h.stream),
                       "sc_granger_pairwise_f64")
            iters, not_conv, fallback = max(iters, summary[0]), not_conv + summary[1], fallback + summary[2]
            n_iter_all.append(np.array(h.download(it_c, (n_groups * n,), np.int32)))
            status_all.append(np.array(h.download(st_c, (n_groups * n,), np.int32)))
            for b in (d_pairs, it_c, st_c):
                b.free()
        res = h.download(out, (n_groups, F, C, C), np.float64)
        work.free(); out.free()
        status = np.concatenate(status_all)
        if fallback:
            logger.warning("Computing the initial conditions using the Cholesky failed. "
                           f"Using the identity as initial condition ({fallback} problems).")
        if not_conv:
            logger.warning(f"Maximum iterations reached. {status.size - not_conv} of {status.size} converged")
        self._last_wilson = dict(iterations=int(iters), not_converged=int(not_conv), cholesky_fallbacks=int(fallback),
                                 n_iter=np.concatenate(n_iter_all), status=status)
        return res.reshape(self._kept_shape() + (F, C, C))

    def _mvar_factor_device(self):
        if getattr(self, "_mvar_G", None) is not None:
            return self._mvar_G
        h, lib = host(), host().lib
        N, C = self._shape5[3], self._shape5[4]
        if C > lib.sc_mvar_max_signals():
            raise ValueError(f"the full Wilson factorisation supports n_signals <= "
                             f"{lib.sc_mvar_max_signals()} (got {C}); use the pairwise measures")
        rec, n_obs, n_freq = self._csm_records("granger")
        n_groups = rec.n_bins // n_freq
        nbytes = ctypes.c_size_t()
        _lib.check(lib.sc_mvar_workspace_bytes(n_groups, C, N, byref(nbytes)), "sc_mvar_workspace_bytes")
        work = h.alloc(nbytes.value)
        G = h.alloc(n_groups * N * C * C * 16)
        n_iter, status = h.alloc(n_groups * 4), h.alloc(n_groups * 4)
        summary = (ctypes.c_int32 * 3)(0, 0, 0)
        _lib.check(lib.sc_mvar_factor_f64(rec.buf.ptr, None, n_groups, n_freq, N, C, rec.planes(_lib.PLANE_CSM),
                                          self._n_observations_total(n_obs), 1e-8, 60, work.ptr, nbytes.value, G.ptr, n_iter.ptr, status.ptr,
                                          summary, h.stream), "sc_mvar_factor_f64")
        st = np.array(h.download(status, (n_groups,), np.int32))
        iters, not_conv, fallback = int(summary[0]), int(summary[1]), int(summary[2])
        if fallback:
            logger.warning("Computing the initial conditions using the Cholesky failed. "
                           f"Using the identity as initial condition ({fallback} windows).")
        if not_conv:
            logger.warning(f"Maximum iterations reached. {st.size - not_conv} of {st.size} converged")
        self._last_wilson = dict(iterations=iters, not_converged=not_conv, cholesky_fallbacks=fallback,
                                 n_iter=np.array(h.download(n_iter, (n_groups,), np.int32)), status=st)
        for b in (n_iter, status):
            b.free()
        self._mvar_G = (G, work, nbytes.value, n_groups)
        return self._mvar_G

    def _mvar(self, which, n_freq_axis=True):
        h = host()
        G, work, nbytes, n_groups = self._mvar_factor_device()
        N, C = self._shape5[3], self._shape5[4]
        F = N // 2 + 1
        cplx = which in (_lib.MVAR_TRANSFER, _lib.MVAR_COEFFICIENTS)
        shape = (n_groups, F, C, C) if n_freq_axis else (n_groups, C, C)
        dt = np.complex128 if cplx else np.float64
        dev = h.alloc(int(np.prod(shape)) * np.dtype(dt).itemsize)
        _lib.check(h.lib.sc_mvar_measure_f64(G.ptr, n_groups, N, C, which, dev.ptr, work.ptr, nbytes, h.stream), "sc_mvar_measure_f64")
        out = h.download(dev, shape, dt)
        dev.free()
        return out.reshape(self._kept_shape() + shape[1:])

    @property
    def _minimum_phase_factor(self):
        h = host()
        G, _, _, n_groups = self._mvar_factor_device()
        N, C = self._shape5[3], self._shape5[4]
        out = np.array(h.download(G, (n_groups, N, C, C), np.complex128))
        return out.reshape(self._kept_shape() + out.shape[1:])

    def global_coherence(self, max_rank=1):
        h, lib = host(), host().lib
        W, R, K, N, C = self._shape5
        max_rank = int(max_rank)
        if not 1 <= max_rank <= min(C, R * K):
            raise ValueError(f"max_rank must be between 1 and min(n_signals, n_trials * n_tapers) = {min(C, R * K)}")
        if C > lib.sc_global_coherence_max_signals():
            raise ValueError(f"global_coherence supports n_signals <= {lib.sc_global_coherence_max_signals()}")
        rec, n_obs, n_freq = self._csm_records("global", "trials_tapers")
        values, vectors = h.alloc(W * N * max_rank * 8), h.alloc(W * N * C * max_rank * 16)
        _lib.check(lib.sc_global_coherence_f64(rec.buf.ptr, W, n_freq, N, C, rec.planes(_lib.PLANE_CSM), self._n_observations_total(n_obs),
                                               max_rank, int(max_rank < C - 1), values.ptr, vectors.ptr, h.stream), "sc_global_coherence_f64")
        res = (np.array(h.download(values, (W, N, max_rank), np.float64)),
               np.array(h.download(vectors, (W, N, C, max_rank), np.complex128)))
        values.free(); vectors.free()
        return res

    def canonical_coherence(self, group_labels):
        h, lib = host(), host().lib
        group_labels = np.asarray(group_labels)
        labels = np.unique(group_labels)
        groups = [np.flatnonzero(np.isin(group_labels, lab)) for lab in labels]
        rec, n_obs, _ = self._csm_records("canonical", "trials_tapers", two_sided=False)
        n_total = self._n_observations_total(n_obs)
        small = [k for k, g in enumerate(groups) if len(g) < n_total]
        max_group = int(lib.sc_canonical_max_group())
        if any(len(groups[k]) > max_group for k in small):
            raise ValueError(f"canonical_coherence: groups of more than {max_group} channels need n_trials * n_tapers "
                             "<= the group size (their coherence is then 1) -- the whitening kernel takes up to "
                             f"{max_group} channels per group")
        n_g, n_bins, C = len(groups), rec.n_bins, self._shape5[4]
        res = np.ones((n_bins, n_g, n_g))
        res[:, np.arange(n_g), np.arange(n_g)] = np.nan
        if len(small) >= 2:
            cmax = max(len(groups[k]) for k in small)
            stride = 16 if cmax <= 16 else (32 if cmax <= 32 else 128)
            members = np.full((len(small), stride), -1, dtype=np.int32)
            for i, k in enumerate(small):
                members[i, :len(groups[k])] = groups[k]
            sizes = np.array([len(groups[k]) for k in small], dtype=np.int32)
            d_members, d_sizes = h.upload(members), h.upload(sizes)
            out, fail = h.alloc(n_bins * len(small) * len(small) * 8), h.alloc(4)
            _lib.check(lib.sc_memset_zero(fail.ptr, 4, h.stream), "sc_memset_zero")
            _lib.check(lib.sc_canonical_coherence_f64(rec.buf.ptr, n_bins, C, rec.planes(_lib.PLANE_CSM), n_total, d_members.ptr,
                                                      d_sizes.ptr, len(small), int(cmax), out.ptr, fail.ptr, h.stream),
                       "sc_canonical_coherence_f64")
            sub = np.array(h.download(out, (n_bins, len(small), len(small)), np.float64))
            n_fail = int(h.download(fail, (1,), np.int32)[0])
            res[np.ix_(np.arange(n_bins), small, small)] = sub
            for b in (d_members, d_sizes, out, fail):
                b.free()
            if n_fail:
                logger.warning(f"{n_fail} group cross-spectral blocks were not positive definite (NaN output)")
        W = self._shape5[0]
        return res.reshape(W, self._n_freq, n_g, n_g), labels
