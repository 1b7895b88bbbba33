"""Connectivity measures: host-side mirror of the reference's ``Connectivity`` API.

Same constructor, properties, method names, output shapes (NumPy float64 / complex128,
non-negative frequencies only) and error behaviour as
``spectral_connectivity.connectivity.Connectivity`` (reference connectivity.py:163-1650) for
the hot-path measures.  All arithmetic runs on an MI355X through ``libsc_hip.so``:

    expectation of the cross-spectral matrix and the per-observation planes |Im s|, (Im s)^2, sign Im s, s/|s|
        float32 engine (dtype=complex64): one pass on the bf16 matrix pipe, bf16x3 split (sc_fused.hip), an f32 VALU kernel
                       for few channels, the f32-MFMA / per-plane VALU kernels for strided spectra (sc_csm.hip, sc_nonlinear.hip)
        float64 engine (dtype=complex128, the default): fp64 matrix cores + fp64 VALU, double records (sc_f64.hip)
    measure algebra, eps clamps, NaN diagonals  -> fp64 epilogue, several measures per launch (sc_measure.hip)
    pairwise Granger / full Wilson + MVAR measures / canonical / global coherence -> sc_wilson.hip, sc_wilson_fft.hip,
        sc_mvar.hip, sc_canonical.hip, sc_global.hip (fp64)

There is no NumPy fallback: without the HIP extension / a GPU every measure raises.
"""
import warnings
from logging import getLogger

import numpy as np

from . import _lib
from ._lib import EXPECTATION_AXES

# scale of the Tikhonov terms in the MVAR quantities (reference connectivity.py: same name and value; applied on the
# device in sc_mvar.hip / sc_wilson.hip)
TIKHONOV_REGULARIZATION_FACTOR = 1e-12

logger = getLogger(__name__)

# kept for API parity with reference connectivity.py:67-75 (keys are what matters)
EXPECTATION = dict.fromkeys(EXPECTATION_AXES)


class _PendingSpectra:
    """A float32-engine transform that has not run yet (Connectivity.from_multitaper): sizes only."""

    def __init__(self, multitaper, precision):
        self.multitaper, self.precision = multitaper, precision
        shape = multitaper.time_series.shape
        self.shape5 = (int(multitaper.n_time_windows), int(shape[1]), int(multitaper.n_tapers),
                       int(multitaper.n_fft_samples), int(shape[2]))


class Connectivity:
    """Frequency-domain connectivity measures computed on an MI355X.

    Parameters mirror reference connectivity.py:277-285.  ``fourier_coefficients`` is either
    a 5-D complex array (n_time_windows, n_trials, n_tapers, n_fft_samples, n_signals) --
    uploaded once -- or, through :meth:`from_multitaper`, the HBM-resident spectra of a
    :class:`~spectral_connectivity_amd.transforms.Multitaper` (no host round trip).

    ``blocks`` is accepted for compatibility and ignored: the engine never materialises the
    per-observation cross-spectra the reference blocks over.

    ``dtype`` selects the device arithmetic, as it selects the arithmetic of the reference's cross-spectral products
    (connectivity.py:277-285, :1799-1822): ``numpy.complex128`` (the default, like the reference) runs the float64
    engine -- float64 transform, cross-spectra on the fp64 matrix cores, float64 measures: results equal to the
    reference's to ~1e-12 --, ``numpy.complex64`` the float32 engine of the headline path (fused f32 transform, bf16x3
    / f32 matrix cores, fp64 epilogue: ~1e-6 on power, ~1e-7 of the array maximum on cross-spectral measures, 2-3x
    faster).  ``options.precision`` can force either engine for the whole process.
    """

    def __init__(self, fourier_coefficients, expectation_type="trials_tapers", frequencies=None,
                 time=None, blocks=None, dtype=np.complex128):
        from . import options
        self._precision = options.engine_precision(dtype)
        self._spectra = None
        self._pending = None
        self._deferred_source = None      # (Multitaper, precision) whose transform left checks pending: _settle
        self._host_coefficients = None
        self._multitaper = None
        if getattr(fourier_coefficients, "is_device_spectra", False):
            self._spectra = fourier_coefficients
        elif isinstance(fourier_coefficients, _PendingSpectra):
            self._pending = fourier_coefficients
        else:
            fourier_coefficients = np.asarray(fourier_coefficients)
            if fourier_coefficients.ndim != 5:
                raise ValueError(
                    f"fourier_coefficients must be 5-dimensional, got {fourier_coefficients.ndim}D array.\n"
                    "Expected shape: (n_time_windows, n_trials, n_tapers, n_fft_samples, n_signals)\n"
                    f"Got shape: {fourier_coefficients.shape}\n\n"
                    "If you have time series data, use the Multitaper class to transform it:\n"
                    "  from spectral_connectivity_amd import Multitaper\n"
                    "  m = Multitaper(time_series, sampling_frequency=your_fs, ...)\n"
                    "  fourier_coefficients = m.fft()")
            self._host_coefficients = fourier_coefficients
        if expectation_type not in EXPECTATION_AXES:
            words = set(expectation_type.split("_"))
            suggestion = None
            if words.issubset({"time", "trials", "tapers"}):
                for key in EXPECTATION_AXES:
                    if set(key.split("_")) == words:
                        suggestion = key
                        break
            msg = (f"Invalid expectation_type '{expectation_type}' is not supported.\n"
                   "This parameter controls which dimensions to average over when computing "
                   "the cross-spectral matrix.\n")
            if suggestion:
                msg += f"\nDid you mean '{suggestion}'? (The words must be in a specific order)\n"
            msg += "\nValid options are:\n" + "".join(f"  - '{k}'\n" for k in sorted(EXPECTATION_AXES))
            msg += "\nMost common: 'trials_tapers' (average over both trials and tapers)"
            raise ValueError(msg)
        if self._host_coefficients is not None and not np.all(np.isfinite(self._host_coefficients)):
            warnings.warn(
                "fourier_coefficients contains NaN or Inf values. This may indicate:\n"
                "  - NaN/Inf in your input time series data\n"
                "  - Issues with windowing parameters (e.g., window too short)\n"
                "  - Numerical instability in preprocessing", UserWarning, stacklevel=2)
        self.expectation_type = expectation_type
        self._frequencies = frequencies
        self._blocks = blocks
        self._dtype = dtype
        self.time = None if time is None else np.asarray(time)
        self._accum_cache = {}

    @classmethod
    def from_multitaper(cls, multitaper_instance, expectation_type="trials_tapers", blocks=None,
                        dtype=np.complex128):
        """Reference connectivity.py:366-400, but the coefficients never leave the device."""
        from . import options
        precision = options.engine_precision(dtype)
        if precision == "float32" and not np.iscomplexobj(multitaper_instance.time_series):
            # float32 engine: the transform runs when the first measure asks for its accumulators -- which families they
            # are decides the device format of the spectra (CSM / |Im s|: f16 pieces for sc_fused2.hip, engine.multitaper_spectra)
            multitaper_instance.check_device_path()
            obj = cls(_PendingSpectra(multitaper_instance, precision), expectation_type=expectation_type,
                      time=multitaper_instance.time, frequencies=multitaper_instance.frequencies, blocks=blocks, dtype=dtype)
        else:
            obj = cls(multitaper_instance.device_spectra(precision=precision),
                      expectation_type=expectation_type, time=multitaper_instance.time,
                      frequencies=multitaper_instance.frequencies, blocks=blocks, dtype=dtype)
        obj._multitaper = multitaper_instance
        return obj

    # ---- bookkeeping ---------------------------------------------------------------------
    @property
    def fourier_coefficients(self):
        if self._host_coefficients is None:
            self._host_coefficients = self._multitaper.fft()
        return self._host_coefficients

    @property
    def _shape5(self):
        if self._host_coefficients is not None:
            return self._host_coefficients.shape
        if self._spectra is None and self._pending is not None:
            return self._pending.shape5
        s = self._spectra
        return (s.W, s.R, s.K, s.n_fft, s.C)

    @property
    def frequencies(self):
        """Non-negative frequencies (reference connectivity.py:402-424)."""
        if self._frequencies is None:
            return None
        f = np.asarray(self._frequencies)
        out = np.array(f[: len(f) // 2 + 1], dtype=float)
        if len(out) and out[-1] < 0:
            out[-1] = abs(out[-1])
        return out

    @property
    def all_frequencies(self):
        return None if self._frequencies is None else np.asarray(self._frequencies)

    @property
    def n_observations(self):
        """Reference connectivity.py:594-610."""
        return int(np.prod([self._shape5[a] for a in EXPECTATION_AXES[self.expectation_type]]))

    # ---- device plumbing -----------------------------------------------------------------
    def _planes_request_ok(self, planes):
        """Would the planes-format stage B (sc_fused2.hip) take ``planes`` with this object's expectation type and shape?  Asked
        BEFORE a pending transform picks the device format of the spectra: its observations must form one run of rows (every
        expectation type but "time_tapers" with several trials)."""
        from ctypes import byref
        from ._lib import SpectraDesc
        W, R, K, N, C = (int(v) for v in self._shape5)
        C_alloc = C + 1 if (C % 2 and C + 1 <= _lib.PLANES_FORMAT_MAX_CHANNELS) else C
        axes = EXPECTATION_AXES[self.expectation_type]
        d = SpectraDesc(n_freq=N // 2 + 1, n_windows=W, n_trials=R, n_tapers=K, n_signals=C_alloc, stride_freq=W * R * K * C_alloc,
                        stride_window=R * K * C_alloc, stride_trial=K * C_alloc, stride_taper=C_alloc, reduce_window=int(0 in axes),
                        reduce_trial=int(1 in axes), reduce_taper=int(2 in axes), reserved=0)
        return bool(_lib.load().sc_fused2_supported(byref(d), planes))

    def _settle(self):
        """Settle what the transform left pending (Multitaper.settle_device_checks: the deferred NaN / infinity warning, the planes
        format's quality check) -- called where a result is downloaded anyway.  False: the spectra were replaced (complex64 instead
        of the planes format) and everything computed from the old ones is dropped; the caller computes again."""
        src = self._deferred_source
        if src is None:
            return True
        self._deferred_source = None
        m, precision = src
        if m.settle_device_checks(precision):
            return True
        self._spectra = m.device_spectra(precision=precision)
        self._accum_cache.clear()
        return False

    def _device(self, planes_hint=None, defer_checks=False):
        """The device spectra; ``planes_hint``: the accumulator families about to be requested.  A pending transform writes the
        planes format when the SHAPE qualifies for it and the hint is one of the families its kernels serve -- whichever
        (_lib.planes_format_applies) -- and this object's expectation type can run on it.  ``defer_checks``: the caller downloads a
        result next and calls _settle() there (and computes again if that says so); every other caller gets settled spectra."""
        if self._spectra is None and self._pending is not None:
            if planes_hint is not None and not (planes_hint in _lib.PLANES_FORMAT_FAMILIES and self._planes_request_ok(planes_hint)):
                planes_hint = None
            m, precision = self._pending.multitaper, self._pending.precision
            self._spectra = m.device_spectra(precision=precision, planes_hint=planes_hint, defer_checks=True)
            self._deferred_source = (m, precision)
            self._pending = None
        if not defer_checks and self._deferred_source is not None:
            self._settle()
        if self._spectra is None:
            from . import engine
            _lib.require_gpu()
            self._spectra = engine.upload_coefficients(self._host_coefficients, f64=self._precision == "float64")
        return self._spectra

    @property
    def _n_freq(self):
        return self._shape5[3] // 2 + 1

    def _accumulators(self, planes, defer_checks=False):
        """Accumulator record containing at least ``planes`` (cached)."""
        from . import engine
        for have, rec in self._accum_cache.items():
            if isinstance(have, int) and have & planes == planes:
                return have, rec
        sp = self._device(planes_hint=planes, defer_checks=defer_checks)
        have = None
        single = self._reduce_over_ranks.__func__ is Connectivity._reduce_over_ranks
        if sp.f64 and single:
            # float64 engine, single process: families a cached record already holds are copied, not recomputed
            best = max((h for h in self._accum_cache if isinstance(h, int) and h & planes), key=lambda h: bin(h & planes).count("1"),
                       default=None)
            if best is not None:
                planes |= best                     # the new record supersedes the old one
                have = (best, self._accum_cache[best][0])
        from . import options
        if (sp.P is not None and planes == _lib.PLANE_CSM and options.anticipate_phase_lag
                and self._planes_request_ok(_lib.PLANE_CSM | _lib.PLANE_ABS_IM)):
            # spectra held as f16 pieces: the launch that sums the cross-spectra also sums |Im s| (options.anticipate_phase_lag) --
            # coherence followed by wPLI, the BASELINE pair, is then ONE pass over the spectra whichever comes first
            planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
        # single process, float32 engine on the planes format: the split-bin partial records stay unfolded (a 3-D tensor) and
        # the epilogue sums them while it reads -- the path bench.py times
        accum, n_obs = engine.accumulate(sp, self.expectation_type, planes, n_freq=self._n_freq, have=have, fold=not single)
        accum = self._reduce_over_ranks(accum)
        if have is not None:
            del self._accum_cache[have[0]]
        for old in [h for h in self._accum_cache if isinstance(h, int) and h & planes == h]:
            del self._accum_cache[old]              # a record the new one covers: its memory can go
        self._accum_cache[planes] = (accum, n_obs)
        return planes, (accum, n_obs)

    def _csm_records(self, tag, expectation_type=None, two_sided=True):
        """CSM records for the consumers that read every bin (Granger, MVAR, global / canonical coherence), cached and
        summed over the trial shards: (records, n_observations of this process, bins per group).  ``two_sided``:
        uploaded coefficients are accumulated on all N bins (real-input spectra hold 0..N/2, mirrored on the device)."""
        from . import engine
        sp = self._device()
        N = self._shape5[3]
        n_freq = (sp.F if sp.real_input else N) if two_sided else self._n_freq
        key = (tag, n_freq)
        if key not in self._accum_cache:
            accum, n_obs = engine.accumulate(sp, expectation_type or self.expectation_type, _lib.PLANE_CSM, n_freq=n_freq)
            self._accum_cache[key] = (self._reduce_over_ranks(accum), n_obs)
        accum, n_obs = self._accum_cache[key]
        return accum, n_obs, n_freq

    def _reduce_over_ranks(self, accum):
        """Sum of the records over the processes that hold the trials: nothing to add here; parallel.ShardedConnectivity
        (one process per GPU, trials sharded) all-reduces."""
        return accum

    def _kept_shape(self):
        W, R, K = self._shape5[:3]
        axes = EXPECTATION_AXES[self.expectation_type]
        return tuple(n for i, n in enumerate((W, R, K)) if i not in axes)

    def _measure(self, which):
        from . import engine
        C = self._shape5[4]
        for _ in range(2):
            # (a fresh transform's two small read-backs are settled with this download, not before stage B: _settle; in the rare
            #  case the planes format is withdrawn there, the measure is computed once more from the complex64 spectra)
            have, (accum, n_obs) = self._accumulators(_lib.MEASURE_PLANES[which], defer_checks=True)
            # the epilogue writes the dtype the reference returns (_wide_output) itself: no widening pass on the way out
            host = engine.to_host(engine.measure(accum, C, have, self._n_observations_total(n_obs), which,
                                                 wide=self._wide_output(which)))
            if self._settle():
                break
        tail = (C,) if which == _lib.M_POWER else (C, C)
        return host.reshape(self._kept_shape() + (self._n_freq,) + tail)

    def _n_observations_total(self, local_n_obs):
        return local_n_obs

    # the measures whose arithmetic the reference runs in `dtype` from the first product on (the per-observation
    # cross-spectra of _complex_inner_product(dtype=...), connectivity.py:1799-1822, then fcn and the expectation)
    _DTYPE_MEASURES = frozenset((_lib.M_PLV, _lib.M_PLV_COMPLEX, _lib.M_PLI, _lib.M_WPLI, _lib.M_DEBIASED_PLI2,
                                 _lib.M_DEBIASED_WPLI2, _lib.M_PPC))

    def _wide_output(self, which):
        """True: float64 / complex128 results, False: float32 / complex64 -- what the reference returns for this
        measure: the phase-lag / phase-locking family comes out in the real type of ``dtype`` (float32 for
        ``dtype=complex64``); power and the coherency family divide by the power, which is computed from the
        coefficients themselves, so they come out in the coefficients' precision (complex128 from ``Multitaper.fft``,
        whatever ``dtype`` says; float32 only for uploaded complex64 coefficients)."""
        if which in self._DTYPE_MEASURES:
            return np.dtype(self._dtype) != np.complex64
        if self._multitaper is None and self._host_coefficients is not None:
            return self._host_coefficients.dtype != np.complex64
        return True

    # accumulator families behind the expectation-type measures of the public interface
    _METHOD_PLANES = {
        "power": _lib.PLANE_CSM, "coherency": _lib.PLANE_CSM, "coherence_phase": _lib.PLANE_CSM,
        "coherence_magnitude": _lib.PLANE_CSM, "imaginary_coherence": _lib.PLANE_CSM,
        "phase_locking_value": _lib.PLANE_UNIT, "pairwise_phase_consistency": _lib.PLANE_UNIT,
        "phase_lag_index": _lib.PLANE_SIGN_IM, "debiased_squared_phase_lag_index": _lib.PLANE_SIGN_IM,
        "weighted_phase_lag_index": _lib.PLANE_CSM | _lib.PLANE_ABS_IM,
        "debiased_squared_weighted_phase_lag_index": _lib.PLANE_CSM | _lib.PLANE_ABS_IM | _lib.PLANE_IM_SQ,
        "phase_slope_index": _lib.PLANE_CSM, "group_delay": _lib.PLANE_CSM, "delay": _lib.PLANE_CSM,
    }

    def _prepare(self, methods):
        """The accumulator families the named measures need, each accumulated ONCE up front instead of a record per measure that
        re-accumulates what it shares with the others.  float32 engine: one record per kernel family -- the cross-spectral one
        (CSM, + |Im s|, + (Im s)^2: one launch, or one + a plane pass), sign(Im s), the unit phasors -- because each family is its
        own pass over the spectra anyway and the planes-format kernels take exactly these sets (a combined CSM + sign record would
        push the whole object back to complex64 spectra); float64 engine: one record (its kernels fill any subset, and a later
        request copies what a record already holds)."""
        planes = 0
        for name in methods:
            planes |= self._METHOD_PLANES.get(name, 0)
        if not planes or bin(planes).count("1") < 2:
            return
        if self._precision == "float64":
            self._accumulators(planes)
            return
        cross = _lib.PLANE_CSM | _lib.PLANE_ABS_IM | _lib.PLANE_IM_SQ
        for family in (planes & cross, planes & _lib.PLANE_SIGN_IM, planes & _lib.PLANE_UNIT):
            if family & (_lib.PLANE_ABS_IM | _lib.PLANE_IM_SQ):
                family |= _lib.PLANE_CSM                       # (the |Im s| / (Im s)^2 planes ride on the cross-spectral launch)
            if family & _lib.PLANE_IM_SQ:
                family |= _lib.PLANE_ABS_IM
            if family and bin(family).count("1") > 1:
                self._accumulators(family)

    # ---- measures (reference connectivity.py:612-1159) -----------------------------------
    def power(self):
        """Power spectral density, non-negative frequencies: (..., n_freq, n_signals)."""
        return self._measure(_lib.M_POWER)

    def _expectation_cross_spectral_matrix(self):
        """E[X_i conj X_j] on the non-negative bins (reference connectivity.py:463-526)."""
        return self._measure(_lib.M_CSM)

    def coherency(self):
        return self._measure(_lib.M_COHERENCY)

    def coherence_phase(self):
        return self._measure(_lib.M_COHERENCE_PHASE)

    def coherence_magnitude(self):
        return self._measure(_lib.M_COHERENCE_MAGNITUDE)

    def imaginary_coherence(self):
        return self._measure(_lib.M_IMAGINARY_COHERENCE)

    def _phase_locking_value(self):
        return self._measure(_lib.M_PLV_COMPLEX)

    def phase_locking_value(self):
        return self._measure(_lib.M_PLV)

    def phase_lag_index(self):
        return self._measure(_lib.M_PLI)

    def weighted_phase_lag_index(self):
        return self._measure(_lib.M_WPLI)

    def debiased_squared_phase_lag_index(self):
        return self._measure(_lib.M_DEBIASED_PLI2)

    def debiased_squared_weighted_phase_lag_index(self):
        return self._measure(_lib.M_DEBIASED_WPLI2)

    def pairwise_phase_consistency(self):
        return self._measure(_lib.M_PPC)

    # ---- pairwise spectral Granger (reference connectivity.py:1161-1213) -----------------
    def _granger(self, pairs):
        from . import engine
        N, C = self._shape5[3], self._shape5[4]
        return engine.to_host(self._granger_device(pairs)).reshape(self._kept_shape() + (N // 2 + 1, C, C))

    def _granger_device(self, pairs):
        """Device tensor [n_groups, N/2+1, C, C] float64: the listed pairs filled in, NaN elsewhere."""
        from . import engine
        N, C = self._shape5[3], self._shape5[4]
        planes = _lib.PLANE_CSM
        accum, n_obs, n_freq = self._csm_records("granger")
        n_groups = accum.shape[0] // n_freq
        out, n_iter, status, (iters, not_conv, fallback) = engine.granger_pairwise(
            accum, n_groups, n_freq, N, C, planes, self._n_observations_total(n_obs), pairs)
        if fallback:
            # reference minimum_phase_decomposition.py:78-93 (there the start is a random draw around the identity)
            logger.warning("Computing the initial conditions using the Cholesky failed. "
                           f"Using the identity as initial condition ({fallback} problems).")
        if not_conv:
            logger.warning(f"Maximum iterations reached. {status.numel() - not_conv} of {status.numel()} converged")
        self._last_wilson = dict(iterations=iters, not_converged=not_conv, cholesky_fallbacks=fallback,
                                 n_iter=n_iter.cpu().numpy(), status=status.cpu().numpy())
        return out

    def pairwise_spectral_granger_prediction(self):
        """Power at node i explained by node j, out[..., i, j] = j -> i (diagonal NaN)."""
        C = self._shape5[4]
        pairs = np.array([(i, j) for i in range(C) for j in range(i + 1, C)], dtype=np.int32)
        if pairs.size == 0:
            return np.full(self._kept_shape() + (self._n_freq, C, C), np.nan)
        return self._granger(pairs)

    def subset_pairwise_spectral_granger_prediction(self, pairs):
        """Granger prediction for the listed (i, j) channel pairs only; every other entry is NaN
        (reference connectivity.py:1193-1213).  Indices follow NumPy rules: negative indices count from the end,
        anything outside [-n_signals, n_signals) raises IndexError."""
        C = self._shape5[4]
        pairs = np.asarray(pairs)
        if pairs.size == 0:
            return np.full(self._kept_shape() + (self._n_freq, C, C), np.nan)
        if pairs.ndim != 2 or pairs.shape[1] != 2:
            raise ValueError(f"pairs must be a sequence of (i, j) index pairs, got shape {pairs.shape}")
        if not np.issubdtype(pairs.dtype, np.integer):
            if not np.all(pairs == np.floor(pairs)):
                raise IndexError("pair indices must be integers")
            pairs = pairs.astype(np.int64)
        bad = (pairs < -C) | (pairs >= C)
        if bad.any():
            raise IndexError(f"index {int(pairs[bad][0])} is out of bounds for axis with size {C}")
        pairs = np.where(pairs < 0, pairs + C, pairs).astype(np.int32)
        # a channel paired with itself is a singular 2 x 2 problem; its only entries lie on the (NaN) diagonal
        pairs = pairs[pairs[:, 0] != pairs[:, 1]]
        if pairs.size == 0:
            return np.full(self._kept_shape() + (self._n_freq, C, C), np.nan)
        return self._granger(pairs)

    # ---- full Wilson factor and the directed MVAR measures (reference connectivity.py:567-589,
    # :1237-1426): one batched C x C factorisation on the device, cached, then one small kernel
    # per measure ------------------------------------------------------------------------------
    def _mvar_factor_device(self):
        from . import engine
        if getattr(self, "_mvar_G", None) is not None:
            return self._mvar_G
        sp = self._device()
        N, C = self._shape5[3], self._shape5[4]
        if C > _lib.load().sc_mvar_max_signals():
            raise ValueError(f"the full Wilson factorisation supports n_signals <= "
                             f"{_lib.load().sc_mvar_max_signals()} (got {C}); use the pairwise measures")
        planes = _lib.PLANE_CSM
        accum, n_obs, n_freq = self._csm_records("granger")
        n_groups = accum.shape[0] // n_freq
        G, n_iter, status, (iters, not_conv, fallback) = engine.mvar_factor(
            n_groups, N, C, accum=accum, n_freq_accum=n_freq, planes=planes, n_obs=self._n_observations_total(n_obs))
        st = status.cpu().numpy()
        if fallback:
            # reference minimum_phase_decomposition.py:78-93 (there the start is a random draw around the identity)
            logger.warning("Computing the initial conditions using the Cholesky failed. "
                           f"Using the identity as initial condition ({fallback} windows).")
        if not_conv:
            logger.warning(f"Maximum iterations reached. {st.size - not_conv} of {st.size} converged")
        self._last_wilson = dict(iterations=iters, not_converged=not_conv, cholesky_fallbacks=fallback,
                                 n_iter=n_iter.cpu().numpy(), status=st)
        self._mvar_G = G
        return G

    def _mvar(self, which, n_freq_axis=True):
        from . import engine
        out = engine.to_host(engine.mvar_measure(self._mvar_factor_device(), which))
        C = self._shape5[4]
        tail = (self._shape5[3] // 2 + 1, C, C) if n_freq_axis else (C, C)
        return out.reshape(self._kept_shape() + tail)

    @property
    def _minimum_phase_factor(self):
        from . import engine
        G = engine.to_host(self._mvar_factor_device())
        return G.reshape(self._kept_shape() + G.shape[1:])

    @property
    def _transfer_function(self):
        return self._mvar(_lib.MVAR_TRANSFER)

    @property
    def _noise_covariance(self):
        return self._mvar(_lib.MVAR_NOISE_COVARIANCE, n_freq_axis=False)

    @property
    def _MVAR_Fourier_coefficients(self):
        return self._mvar(_lib.MVAR_COEFFICIENTS)

    def directed_transfer_function(self):
        """|H_ij|^2 normalised by the total inflow into node i; out[..., i, j] = j -> i, range [0, 1]."""
        return self._mvar(_lib.MVAR_DTF)

    def directed_coherence(self):
        """Transfer-function coupling scaled by the noise variance, normalised by the inflow."""
        return self._mvar(_lib.MVAR_DC)

    def partial_directed_coherence(self, keep_cupy=False):
        """|A_ij|^2 of the MVAR Fourier coefficients normalised by the total outflow of node j."""
        return self._mvar(_lib.MVAR_PDC)

    def generalized_partial_directed_coherence(self):
        """Partial directed coherence with every row scaled by its noise variance."""
        return self._mvar(_lib.MVAR_GPDC)

    def direct_directed_transfer_function(self):
        """Full-frequency directed transfer function times sqrt(partial directed coherence)."""
        return self._mvar(_lib.MVAR_DDTF)

    # ---- band statistics of the coherency (reference connectivity.py:1428-1650): host-side post-processing
    # of the device coherency, see _postprocess.py ----------------------------------------------------
    def phase_slope_index(self, frequencies_of_interest=None, frequency_resolution=None):
        """Weighted average of the coherency phase slope projected on the imaginary axis (Nolte et al. 2008);
        out[..., i, j] > 0 when i leads j.  Shape (..., n_signals, n_signals)."""
        from . import _postprocess as pp
        return pp.phase_slope_index(self.coherency(), self.frequencies, frequencies_of_interest, frequency_resolution)

    def group_delay(self, frequencies_of_interest=None, frequency_resolution=None, significance_threshold=0.05):
        """Average time delay of a broadband signal between every pair: (delay, slope, r_value) from the linear
        regression of the unwrapped coherence phase on frequency over the significant bins of the band."""
        from . import _postprocess as pp
        return pp.group_delay(self.coherency(), self.frequencies, self.n_observations, frequencies_of_interest,
                              frequency_resolution, significance_threshold)

    def delay(self, frequencies_of_interest=None, frequency_resolution=None, significance_threshold=0.05, n_range=3):
        """Range of possible delays (2 pi ambiguity of the coherence phase) per band frequency and pair."""
        from . import _postprocess as pp
        return pp.delay(self.coherency(), self.frequencies, self.n_observations, frequencies_of_interest,
                        frequency_resolution, significance_threshold, n_range)

    # ---- global coherence (reference connectivity.py:822-895) ------------------------------
    def global_coherence(self, max_rank=1):
        """Leading squared singular values (/ n_estimates) and left singular vectors of the signals x
        (trials * tapers) coefficient matrix, per time window and (two-sided) frequency bin.

        Returns (values (n_time_windows, n_fft_samples, max_rank), vectors (n_time_windows,
        n_fft_samples, n_signals, max_rank)).  Computed as the leading eigenpairs of the cross-spectral
        matrix that is already on the device.  Like the reference, the max_rank largest values come
        smallest-first when max_rank < n_signals - 1 (scipy svds) and largest-first otherwise; vectors
        are unit norm with their largest component real positive (the reference's phase is arbitrary).
        """
        from . import engine
        sp = self._device()
        W, R, K, N, C = self._shape5
        max_rank = int(max_rank)
        if not 1 <= max_rank <= min(C, R * K):
            raise ValueError(f"max_rank must be between 1 and min(n_signals, n_trials * n_tapers) = {min(C, R * K)}")
        if C > _lib.load().sc_global_coherence_max_signals():
            raise ValueError(f"global_coherence supports n_signals <= {_lib.load().sc_global_coherence_max_signals()}")
        planes = _lib.PLANE_CSM
        accum, n_obs, n_freq = self._csm_records("global", "trials_tapers")
        values, vectors = engine.global_coherence(accum, W, n_freq, N, C, planes, self._n_observations_total(n_obs),
                                                  max_rank, ascending=max_rank < C - 1)
        return values.cpu().numpy(), vectors.cpu().numpy()

    # ---- canonical coherence (reference connectivity.py:745-820) --------------------------
    def canonical_coherence(self, group_labels):
        """Maximal coherence between linear combinations of each pair of channel groups.

        Returns (array (n_time_windows, n_frequencies, n_groups, n_groups), sorted labels).
        Like the reference this always averages over trials and tapers.
        """
        from . import engine
        group_labels = np.asarray(group_labels)
        labels = np.unique(group_labels)
        groups = [np.flatnonzero(np.isin(group_labels, lab)) for lab in labels]
        planes = _lib.PLANE_CSM
        accum, n_obs, _ = self._csm_records("canonical", "trials_tapers", two_sided=False)
        n_total = self._n_observations_total(n_obs)
        # A group with at least as many channels as there are observations spans the whole observation space: the
        # orthonormal row-space basis V_g the reference gets from its thin SVD (connectivity.py:1979-2032) is then the full
        # n_obs-dimensional space, V_g^H V_h has orthonormal columns for ANY other group h, and every singular value of the
        # pair is 1 -- the canonical coherence is 1 (the reference returns 1 +- 3e-15 there: generic, full-row-rank data).
        # Such groups have no Cholesky factor (their cross-spectral block is rank deficient), so they stay out of the
        # device kernel; the pairs among the remaining groups go through the CSM form as before.
        small = [k for k, g in enumerate(groups) if len(g) < n_total]
        max_group = int(_lib.load().sc_canonical_max_group())
        if any(len(groups[k]) > max_group for k in small):
            raise ValueError(f"canonical_coherence: groups of more than {max_group} channels need n_trials * n_tapers "
                             "<= the group size (their coherence is then 1) -- the whitening kernel takes up to "
                             f"{max_group} channels per group")
        lo, hi, per = self._canonical_bins(accum.shape[0])
        n_g = len(groups)
        import torch
        if hi > lo:
            out = torch.ones((hi - lo, n_g, n_g), dtype=torch.float64, device=accum.device)
            out[:, torch.arange(n_g), torch.arange(n_g)] = float("nan")
            n_fail = 0
            if len(small) >= 2:
                sub, n_fail = engine.canonical_coherence(accum[lo:hi], self._shape5[4], planes, n_total,
                                                         [groups[k] for k in small])
                idx = torch.as_tensor(small, device=accum.device)
                out[:, idx[:, None], idx[None, :]] = sub
        else:                                  # more processes than bins: this one has nothing to evaluate
            out, n_fail = torch.empty((0, n_g, n_g), dtype=torch.float64, device=accum.device), 0
        out = self._canonical_gather(out, accum.shape[0], per)
        if n_fail:
            logger.warning(f"{n_fail} group cross-spectral blocks were not positive definite (NaN output)")
        W = self._shape5[0]
        return out.cpu().numpy().reshape(W, self._n_freq, len(labels), len(labels)), labels

    def _canonical_bins(self, n_bins):
        """Bins [lo, hi) this process evaluates and the per-process count (all of them here; 1/N of them in
        parallel.ShardedConnectivity)."""
        return 0, n_bins, n_bins

    def _canonical_gather(self, part, n_bins, per):
        return part

    def conditional_spectral_granger_prediction(self):
        raise NotImplementedError   # reference connectivity.py:1215-1224 raises too

    def blockwise_spectral_granger_prediction(self):
        raise NotImplementedError   # reference connectivity.py:1226-1235 raises too
