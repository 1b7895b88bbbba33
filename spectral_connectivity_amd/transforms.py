"""Multitaper transform: host-side mirror of the reference's ``Multitaper`` API.

Same constructor, properties and error/warning behaviour as
``spectral_connectivity.transforms.Multitaper`` (reference transforms.py:442-1171), but the
work -- window extraction, detrending, taper multiply (hand-written HIP) and the batched
real FFT (rocFFT) -- runs on an MI355X through ``libsc_hip.so`` and the coefficients stay
resident in HBM for ``Connectivity.from_multitaper``.  Host code here is parameter logic
and the (tiny, once-per-object, float64) DPSS taper generation only.
"""
import os
import warnings
from logging import getLogger
from typing import TypedDict

import numpy as np
from scipy.fft import fft as _host_fft
from scipy.fft import fftfreq, ifft as _host_ifft, next_fast_len
from scipy.linalg import eigh_tridiagonal

logger = getLogger(__name__)

# The reference's backend plug point (transforms.py:405-439): SPECTRAL_CONNECTIVITY_ENABLE_GPU == "true" asks for the
# GPU backend at import time and a missing backend is a RuntimeError right here.  This package has one backend, the
# HIP engine: "true" loads libsc_hip.so now (RuntimeError if that fails), unset loads it on first use, any other value
# (the reference's NumPy path) is refused at the first computation (_lib.require_gpu).
if os.environ.get("SPECTRAL_CONNECTIVITY_ENABLE_GPU") == "true":
    from . import _lib as _hip_lib
    _hip_lib.honour_gpu_switch()
    logger.info("Using GPU for spectral_connectivity: HIP engine " + _hip_lib.library_path())
else:
    logger.info("spectral_connectivity_amd: the HIP engine loads on first use (no CPU backend)")

MIN_EIGENVALUE_THRESHOLD = 0.9   # reference transforms.py:22
TAPER_MULTIPLIER = 2.0           # reference transforms.py:30


def estimate_frequency_resolution(sampling_frequency, time_window_duration, time_halfbandwidth_product):
    """Frequency resolution 2*NW/T in Hz (reference transforms.py:63-141)."""
    return TAPER_MULTIPLIER * time_halfbandwidth_product / time_window_duration


def estimate_n_tapers(time_halfbandwidth_product):
    """floor(2*NW) - 1 (reference transforms.py:144-196)."""
    return int(np.floor(TAPER_MULTIPLIER * time_halfbandwidth_product)) - 1


class MultitaperParameters(TypedDict):
    """Parameter suggestion returned by :func:`suggest_parameters` (reference transforms.py:33-60)."""

    sampling_frequency: float
    time_halfbandwidth_product: float
    time_window_duration: float
    n_tapers: int
    frequency_resolution: float
    n_time_windows: int
    nyquist_frequency: float


def suggest_parameters(sampling_frequency, signal_duration, desired_freq_resolution=None,
                       desired_n_tapers=None):
    """Suggest multitaper parameters for a recording (same rules as reference transforms.py:199-402).

    No target: NW = 3 and a window of a fifth of the signal (at least 0.5 s, at most the signal).
    ``desired_freq_resolution``: window T = 2 NW / df with NW = 3; impossible if T exceeds the signal;
    if fewer than 3 windows would fit, the window is capped at a third of the signal and NW is lowered
    to df T / 2 (never below 1).  ``desired_n_tapers``: NW = (n + 1) / 2 with the default window.
    Both given: the resolution wins (with a ``UserWarning``).
    """
    if desired_freq_resolution is not None and desired_n_tapers is not None:
        warnings.warn(
            "Both 'desired_freq_resolution' and 'desired_n_tapers' were specified. "
            "This is typically not recommended as they have competing effects on the analysis. "
            "Using 'desired_freq_resolution' and ignoring 'desired_n_tapers'.", UserWarning, stacklevel=2)
        desired_n_tapers = None

    def default_window():
        return min(max(signal_duration / 5.0, 0.5), signal_duration)

    if desired_freq_resolution is not None:
        nw = 3.0
        window = TAPER_MULTIPLIER * nw / desired_freq_resolution
        if window > signal_duration:
            raise ValueError(
                f"Cannot achieve desired frequency resolution of {desired_freq_resolution} Hz "
                f"with signal duration of {signal_duration}s.\n"
                f"Required window duration: {window:.2f}s; available: {signal_duration:.2f}s.\n"
                f"Use a longer signal or a coarser resolution (at least "
                f"{TAPER_MULTIPLIER * nw / signal_duration:.2f} Hz).")
        if window > signal_duration / 3:
            window = signal_duration / 3
            nw = max(desired_freq_resolution * window / 2.0, 1.0)
    elif desired_n_tapers is not None:
        nw = (desired_n_tapers + 1) / 2.0
        window = default_window()
    else:
        nw = 3.0
        window = default_window()
    return {
        "sampling_frequency": sampling_frequency,
        "time_halfbandwidth_product": nw,
        "time_window_duration": window,
        "n_tapers": estimate_n_tapers(nw),
        "frequency_resolution": estimate_frequency_resolution(sampling_frequency, window, nw),
        "n_time_windows": int(np.floor(signal_duration / window)),
        "nyquist_frequency": sampling_frequency / 2.0,
    }


def prepare_time_series(time_series, axis=None):
    """Reshape 1-D / 2-D input to (n_time, n_trials, n_signals) (reference transforms.py:1174-1297)."""
    arr = np.asarray(time_series)
    if arr.ndim == 1:
        return arr[:, np.newaxis, np.newaxis]
    if arr.ndim == 2:
        if axis is None:
            raise ValueError(
                "For 2D input, you must specify the 'axis' parameter.\n"
                f"Input shape: {arr.shape}\n"
                "  - axis='signals' if shape is (n_time_samples, n_signals)\n"
                "  - axis='trials' if shape is (n_time_samples, n_trials)")
        if axis == "signals":
            return arr[:, np.newaxis, :]
        if axis == "trials":
            return arr[:, :, np.newaxis]
        raise ValueError(f"axis must be either 'signals' or 'trials', got: {axis!r}")
    if arr.ndim == 3:
        return arr
    raise ValueError(f"Expected 1D, 2D, or 3D array, got {arr.ndim}D array with shape {arr.shape}")


# ------------------------------------------------------------------------------ DPSS (host)
def tridisolve(d, e, b, overwrite_b=True):
    """Solve the symmetric tridiagonal system (diagonal ``d``, off-diagonal ``e``) for right-hand side ``b``
    (reference transforms.py:1443-1489; LAPACK's banded solver here).  ``overwrite_b`` keeps the reference's
    contract: the solution is written into ``b`` (and returned)."""
    from scipy.linalg import solve_banded
    d, e = np.asarray(d, dtype=float), np.asarray(e, dtype=float)
    band = np.zeros((3, d.shape[0]))
    band[0, 1:], band[1], band[2, :-1] = e, d, e
    x = solve_banded((1, 1), band, np.asarray(b, dtype=float))
    if overwrite_b:
        b[...] = x
        return b
    return x


def tridi_inverse_iteration(d, e, w, x0=None, rtol=1e-8):
    """Eigenvector of the symmetric tridiagonal matrix (d, e) for the eigenvalue closest to ``w`` by inverse
    iteration, normalised to unit length, sign unspecified (reference transforms.py:1492-1536).  ``dpss_windows``
    here takes its eigenvectors from LAPACK instead; this helper is kept for callers of the reference's name."""
    d = np.asarray(d, dtype=float)
    shifted = d - w
    x = np.random.randn(d.shape[0]) if x0 is None else np.asarray(x0, dtype=float)
    x = x / np.linalg.norm(x)
    previous = np.zeros_like(x)
    for _ in range(200):
        if np.linalg.norm(np.abs(x) - np.abs(previous)) <= rtol:
            break
        previous = x
        x = tridisolve(shifted, e, x.copy())
        x = x / np.linalg.norm(x)
    return x


def detrend(data, axis=-1, type="linear", bp=0, overwrite_data=False):
    """Remove a constant or a (piecewise) linear trend along ``axis`` -- the host-side helper the reference exposes
    (reference transforms.py:1798-1915, itself scipy.signal.detrend).  The device path detrends inside the fused
    FFT kernel; this function is for callers that use it directly.

    ``type``: 'constant' / 'c' subtracts the mean; 'linear' / 'l' subtracts the least-squares line of every segment
    between the breakpoints ``bp`` (fitted on the abscissa (1..n)/n like the reference).
    """
    if type not in ("linear", "l", "constant", "c"):
        raise ValueError(
            f"Invalid trend type '{type}' is not supported.\n"
            f"The detrend function only supports linear and constant detrending.\n"
            f"Valid options are:\n"
            f"  - 'linear' or 'l': Remove linear trend (best-fit line)\n"
            f"  - 'constant' or 'c': Remove mean (DC offset)\n"
            f"Example: detrend(data, type='linear')")
    data = np.asarray(data)
    if data.dtype.char not in "dfDF":
        data = data.astype(np.float64)
    if type in ("constant", "c"):
        return data - np.mean(data, axis, keepdims=True)
    n = data.shape[axis]
    edges = np.unique(np.r_[0, bp, n])
    if np.any(edges > n):
        shown = [bp] if isinstance(bp, (int, np.integer)) else list(np.asarray(bp).tolist())
        raise ValueError(
            f"Breakpoint value(s) {edges[edges > n].tolist()} exceed data length.\n"
            f"Data has {n} samples along axis {axis}, but breakpoint(s) are beyond this range.\n"
            f"Breakpoints must be in the range [0, {n}).\n"
            f"Check your breakpoint array: {shown}")
    out = np.moveaxis(data if overwrite_data else data.copy(), axis, 0)
    for lo, hi in zip(edges[:-1], edges[1:]):
        seg = out[lo:hi]
        m = hi - lo
        t = (np.arange(1, m + 1) / m).reshape((m,) + (1,) * (seg.ndim - 1))
        tc = t - t.mean()
        slope = (tc * seg).sum(axis=0) / (tc * tc).sum() if m > 1 else np.zeros(seg.shape[1:], dtype=seg.dtype)
        seg -= seg.mean(axis=0) + slope * tc
    return np.moveaxis(out, 0, axis)


def dpss_windows(n_time_samples_per_window, time_halfbandwidth_product, n_tapers, is_low_bias=True,
                 interp_from=None, interp_kind="linear"):
    """Discrete prolate spheroidal sequences and their concentration eigenvalues.

    Same definition and conventions as reference transforms.py:1539-1613: eigenvectors of
    the Slepian tridiagonal matrix (:1654-1714) for the ``n_tapers`` largest eigenvalues,
    unit l2 norm; symmetric tapers have positive mean, antisymmetric ones start with a
    positive lobe (:1717-1745); concentration by the autocorrelation method (:1768-1795);
    ``is_low_bias`` keeps eigenvalue > 0.9, or the best one if none (:1758-1765).
    The eigenvectors come from LAPACK (stemr) instead of the reference's Python inverse
    iteration.  ``interp_from``: compute the tapers at that (shorter) length and interpolate them to
    ``n_time_samples_per_window`` samples (scipy ``interp1d`` of kind ``interp_kind``), renormalised
    (:1615-1651).  Returns (tapers (K', L), eigenvalues (K',)).
    """
    L = int(n_time_samples_per_window)
    K = int(n_tapers)
    half_bw = float(time_halfbandwidth_product) / L
    t = np.arange(L, dtype=np.float64)
    if interp_from is not None:
        from scipy import interpolate
        small, _ = dpss_windows(int(interp_from), time_halfbandwidth_product, K, is_low_bias=False)
        grid = np.linspace(0, small.shape[-1] - 1, L, endpoint=False)
        tapers = np.stack([interpolate.interp1d(np.arange(small.shape[-1]), row, kind=interp_kind)(grid) for row in small])
    else:
        diag = ((L - 1 - 2 * t) / 2.0) ** 2 * np.cos(2 * np.pi * half_bw)
        off = t[1:] * (L - t[1:]) / 2.0
        if L == 1:
            vecs = np.ones((1, 1))
        else:
            lo = max(L - K, 0)
            _, vecs = eigh_tridiagonal(diag, off, select="i", select_range=(lo, L - 1))
        tapers = np.ascontiguousarray(vecs[:, ::-1].T)            # largest eigenvalue first
    tapers /= np.linalg.norm(tapers, axis=1, keepdims=True)
    # sign conventions
    neg = tapers[::2].sum(axis=1) < 0
    tapers[::2][neg] *= -1
    if tapers.shape[0] > 1:
        peak = np.argmax(np.abs(tapers[1::2, : L // 2]), axis=1)
        for i, p in enumerate(peak):
            if tapers[2 * i + 1, :p].sum() < 0:
                tapers[2 * i + 1] *= -1
    # concentration eigenvalues
    nfft = next_fast_len(2 * L - 1)
    spec = _host_fft(tapers, nfft, axis=-1)
    acorr = np.real(_host_ifft(spec * spec.conj(), axis=-1))[:, :L]
    kernel = 4 * half_bw * np.sinc(2 * half_bw * t)
    kernel[0] = 2 * half_bw
    eig = acorr @ kernel
    if is_low_bias:
        keep = eig > MIN_EIGENVALUE_THRESHOLD
        if not keep.any():
            logger.warning("Could not properly use low_bias, keeping lowest-bias taper")
            keep = np.zeros_like(keep)
            keep[np.argmax(eig)] = True
        tapers, eig = tapers[keep], eig[keep]
    return tapers, eig


def _make_tapers(n_time_samples_per_window, sampling_frequency, time_halfbandwidth_product, n_tapers,
                 is_low_bias=True):
    """(L, K') tapers scaled by sqrt(fs) (reference transforms.py:1408-1440).  The DPSS sequences depend on the window
    geometry alone (1.8 ms of LAPACK at 256 samples, 14 ms at 4096: more than the device pipeline of a small request), so the last
    few geometries are remembered -- like an FFT plan -- and every caller gets its own copy."""
    key = (int(n_time_samples_per_window), float(time_halfbandwidth_product), int(n_tapers), bool(is_low_bias))
    hit = _taper_memo.get(key)
    if hit is None:
        tapers, _ = dpss_windows(n_time_samples_per_window, time_halfbandwidth_product, n_tapers,
                                 is_low_bias=is_low_bias)
        hit = np.ascontiguousarray(tapers.T)
        if len(_taper_memo) >= 8:
            _taper_memo.pop(next(iter(_taper_memo)))
        _taper_memo[key] = hit
    return hit * np.sqrt(sampling_frequency)


_taper_memo = {}


def _n_windows(n_time, window, step):
    """reference transforms.py:1363-1365 (floating-point floor)."""
    return int(np.floor((n_time / step) - (window / step) + 1))


_SHAPE_HELP_1D = (
    "For a single time series, use:\n"
    "  >>> from spectral_connectivity_amd.transforms import prepare_time_series\n"
    "  >>> time_series_3d = prepare_time_series(time_series)\n"
    "Or manually:\n"
    "  >>> time_series_3d = time_series[:, np.newaxis, np.newaxis]")
_SHAPE_HELP_2D = (
    "For 2D data, you must clarify the meaning of the second dimension.\n"
    "Use prepare_time_series() helper (axis='signals' or axis='trials'),\n"
    "or add the missing axis manually with np.newaxis.")


_NONFINITE_WARNING = ("Input time_series contains NaN or infinite values.\n"
                      "This will produce invalid spectral estimates.")


class _DeviceSeries:
    """A time series that already lives in HBM (a ``torch`` tensor on a ROCm device handed to ``Multitaper``): the sizes and
    the NumPy dtype the host-side parameter logic asks for, without a copy; ``numpy.asarray`` of it downloads (the reference's
    ``time_series`` attribute is a host array -- whoever reads it as one gets one)."""

    def __init__(self, tensor):
        self.tensor = tensor
        self.shape = tuple(int(n) for n in tensor.shape)
        self.ndim = tensor.dim()
        self.size = int(tensor.numel())
        self.dtype = np.dtype(str(tensor.dtype).replace("torch.", ""))

    def __array__(self, dtype=None, copy=None):
        a = self.tensor.detach().cpu().numpy()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __len__(self):
        return self.shape[0]


def _as_series(time_series):
    """``numpy.asarray`` for host input; a device tensor (torch, on the GPU) stays where it is."""
    if type(time_series).__module__.startswith("torch") and getattr(time_series, "is_cuda", False):
        return _DeviceSeries(time_series)
    return np.asarray(time_series)


class Multitaper:
    """Multitaper spectral transform on an MI355X (drop-in for the reference class).

    Parameters are those of reference transforms.py:574-589.  ``time_series`` has shape
    (n_time_samples, n_trials, n_signals).

    One difference in WHEN a warning appears: the reference scans the series for NaN / infinity in the constructor
    (transforms.py:746-753).  For real series of 4 M samples or more this class defers that scan to the device, next to
    the upload, and the same ``UserWarning`` is raised by the first transform (``fft()``, or the first measure of a
    ``Connectivity`` built from this object) -- not at all if no transform ever runs.  ``options.finite_check = "host"``
    (or ``SC_HIP_FINITE_CHECK=host``) restores the constructor-time scan for every size.

    Beyond the reference: ``time_series`` may be a ``torch`` tensor that already lives on the GPU (float32 / float64,
    (n_time_samples, n_trials, n_signals)); it is used in place -- no host round trip -- and its NaN / infinity scan runs on
    the device with the first transform.
    """

    def __init__(self, time_series, sampling_frequency=1000, time_halfbandwidth_product=3,
                 detrend_type="constant", time_window_duration=None, time_window_step=None,
                 n_tapers=None, tapers=None, start_time=0, n_fft_samples=None,
                 n_time_samples_per_window=None, n_time_samples_per_step=None, is_low_bias=True):
        self.time_series = _as_series(time_series)
        nd = self.time_series.ndim
        if nd != 3:
            msg = (f"Expected 3D array with shape (n_time_samples, n_trials, n_signals), "
                   f"but got {nd}D array with shape {self.time_series.shape}.\n\n")
            if nd == 1:
                msg += _SHAPE_HELP_1D
            elif nd == 2:
                msg += _SHAPE_HELP_2D
            else:
                msg += (f"Arrays with {nd} dimensions are not supported.\n"
                        "Expected shape: (n_time_samples, n_trials, n_signals)")
            raise ValueError(msg)
        if sampling_frequency <= 0:
            raise ValueError(f"sampling_frequency must be positive, got {sampling_frequency}.\n"
                             "The sampling frequency is the rate at which your data was collected.")
        if time_halfbandwidth_product < 1:
            raise ValueError(
                f"time_halfbandwidth_product must be at least 1, got {time_halfbandwidth_product}.\n"
                "It controls the spectral concentration and the number of tapers; typical values are 1-5.")
        if time_halfbandwidth_product > 10:
            warnings.warn(
                f"time_halfbandwidth_product = {time_halfbandwidth_product} is unusually large.\n"
                "Values above 10 apply very heavy spectral smoothing and are rarely used.",
                UserWarning, stacklevel=2)
        if time_window_duration is not None and time_window_duration <= 0:
            raise ValueError(f"time_window_duration must be positive, got {time_window_duration}.\n"
                             "Use None (default) to analyze the entire time series without windowing.")
        if time_window_step is not None and time_window_step <= 0:
            raise ValueError(f"time_window_step must be positive, got {time_window_step}.\n"
                             "Use None (default) to match time_window_duration (no overlap).")
        if (time_window_step is not None and time_window_duration is not None
                and time_window_step > time_window_duration):
            warnings.warn(
                f"time_window_step ({time_window_step}s) is larger than time_window_duration "
                f"({time_window_duration}s).\nThis creates gaps between analysis windows - some data "
                "will not be analyzed.", UserWarning, stacklevel=2)
        n_time, _, n_signals = self.time_series.shape
        if n_time < n_signals:
            warnings.warn(
                f"Your time series has only {n_time} time points but {n_signals} signals. "
                "This seems unusual and your data may be transposed.\n"
                "Expected shape: (n_time_samples, n_trials, n_signals)", UserWarning, stacklevel=2)
        # The reference scans the whole series here (transforms.py:746-753).  Small series: the same scan on the host (a
        # finite sum proves every sample finite -- NaN and inf propagate -- in one pass without a boolean temporary; only a
        # non-finite sum, which overflow could also produce, needs the element-wise check).  Large real series (the scan
        # of cfg3's 131 M samples costs 18 ms of one core, more than the whole device pipeline): the scan runs on the
        # DEVICE next to the upload (sc_nonfinite_f32 / _f64, one read at HBM rate) and the same warning is raised by
        # the first transform; options.finite_check = "host" keeps the constructor-time scan for every size.
        from . import options as _options
        self._finite_checked = True
        if isinstance(self.time_series, _DeviceSeries):
            if self.time_series.dtype.kind not in "f":
                raise TypeError(f"a device time series must be float32 or float64, got {self.time_series.dtype}")
            self._finite_checked = False                  # scanned on the device, with the first transform
        elif (_options.finite_check == "device" and self.time_series.dtype.kind == "f"
                and self.time_series.size >= _options.FINITE_CHECK_DEVICE_MIN):
            self._finite_checked = False
        elif not (self.time_series.dtype.kind == "f" and self.time_series.size and np.isfinite(self.time_series.sum())) \
                and not np.all(np.isfinite(self.time_series)):
            warnings.warn(_NONFINITE_WARNING, UserWarning, stacklevel=2)

        self.sampling_frequency = sampling_frequency
        self.time_halfbandwidth_product = time_halfbandwidth_product
        self.detrend_type = detrend_type
        self._time_window_duration = time_window_duration
        self._time_window_step = time_window_step
        self.is_low_bias = is_low_bias
        self.start_time = np.asarray(start_time)
        self._n_fft_samples = n_fft_samples
        self._tapers = tapers
        self._n_tapers = n_tapers
        self._n_time_samples_per_window = n_time_samples_per_window
        self._n_samples_per_time_step = n_time_samples_per_step
        self._device_spectra = None
        self._deferred_checks = None      # {precision: (flag, quality, taper norm, redo)}: settle_device_checks

    def __repr__(self):
        return ("Multitaper("
                f"sampling_frequency={self.sampling_frequency!r}, "
                f"time_halfbandwidth_product={self.time_halfbandwidth_product!r},\n"
                f"           time_window_duration={self.time_window_duration!r}, "
                f"time_window_step={self.time_window_step!r},\n"
                f"           detrend_type={self.detrend_type!r}, "
                f"start_time={self.start_time}, n_tapers={self.n_tapers})")

    def summarize_parameters(self):
        """Human-readable summary of the analysis configuration (reference transforms.py:810-923)."""
        n_time = self.time_series.shape[0]
        if self.time_window_step == self.time_window_duration:
            overlap = "(non-overlapping)"
        else:
            pct = 100 * (self.time_window_duration - self.time_window_step) / self.time_window_duration
            overlap = f"({pct:.0f}% overlap)"
        n_windows = int(np.floor((n_time - self.n_time_samples_per_window) / self.n_time_samples_per_step) + 1)
        lines = [
            "Multitaper Spectral Analysis Configuration",
            "===========================================",
            "",
            "Data Shape",
            "----------",
            f"Time samples:    {n_time} ({n_time / self.sampling_frequency:.2f} seconds)",
            f"Signals:         {self.n_signals}",
            f"Trials:          {self.n_trials}",
            "",
            "Spectral Parameters",
            "-------------------",
            f"Sampling frequency:            {self.sampling_frequency} Hz",
            f"Time-halfbandwidth product:    {self.time_halfbandwidth_product}",
            f"Number of tapers:              {self.n_tapers}",
            "",
            "Time Windowing",
            "--------------",
            f"Window duration:  {self.time_window_duration:.3f} s ({self.n_time_samples_per_window} samples)",
            f"Window step:      {self.time_window_step:.3f} s {overlap}",
            f"Number of windows: {n_windows}",
            "",
            "Frequency Analysis",
            "------------------",
            f"Frequency resolution: {self.frequency_resolution:.1f} Hz",
            f"Nyquist frequency:    {self.nyquist_frequency:.1f} Hz",
            f"Frequency range:      0.0 - {self.nyquist_frequency:.1f} Hz",
            f"FFT samples:          {self.n_fft_samples}",
        ]
        return "\n".join(lines)

    # ---- derived parameters (reference transforms.py:925-1145) ---------------------------
    @property
    def tapers(self):
        """(n_time_samples_per_window, n_tapers) tapers, scaled by sqrt(fs)."""
        if self._tapers is None:
            self._tapers = _make_tapers(self.n_time_samples_per_window, self.sampling_frequency,
                                        self.time_halfbandwidth_product, self.n_tapers,
                                        is_low_bias=self.is_low_bias)
        return self._tapers

    @property
    def time_window_duration(self):
        if self._time_window_duration is None:
            self._time_window_duration = self.n_time_samples_per_window / self.sampling_frequency
        return self._time_window_duration

    @property
    def time_window_step(self):
        if self._time_window_step is None:
            self._time_window_step = self.n_time_samples_per_step / self.sampling_frequency
        return self._time_window_step

    @property
    def n_tapers(self):
        if self._n_tapers is None:
            return int(np.floor(TAPER_MULTIPLIER * self.time_halfbandwidth_product - 1))
        return self._n_tapers

    @property
    def n_time_samples_per_window(self):
        if self._n_time_samples_per_window is None and self._time_window_duration is None:
            self._n_time_samples_per_window = self.time_series.shape[0]
        elif self._time_window_duration is not None:
            self._n_time_samples_per_window = int(
                np.around(self.time_window_duration * self.sampling_frequency))
        return self._n_time_samples_per_window

    @property
    def n_fft_samples(self):
        if self._n_fft_samples is None:
            self._n_fft_samples = next_fast_len(self.n_time_samples_per_window)
        return self._n_fft_samples

    @property
    def frequencies(self):
        """Two-sided FFT bin frequencies (reference transforms.py:1038-1048)."""
        return fftfreq(self.n_fft_samples, 1.0 / self.sampling_frequency)

    @property
    def n_time_samples_per_step(self):
        if self._n_samples_per_time_step is None and self._time_window_step is None:
            self._n_samples_per_time_step = self.n_time_samples_per_window
        elif self._time_window_step is not None:
            # truncation, not rounding: reference transforms.py:1068-1070
            self._n_samples_per_time_step = int(self.time_window_step * self.sampling_frequency)
        return self._n_samples_per_time_step

    @property
    def n_time_windows(self):
        return _n_windows(self.time_series.shape[0], self.n_time_samples_per_window,
                          self.n_time_samples_per_step)

    @property
    def time(self):
        """Start time of every window (reference transforms.py:1075-1091)."""
        starts = np.arange(self.n_time_windows) * self.n_time_samples_per_step
        return self.start_time + starts / self.sampling_frequency

    @property
    def n_signals(self):
        return self.time_series.shape[-1]

    @property
    def n_trials(self):
        return self.time_series.shape[1]

    @property
    def frequency_resolution(self):
        return TAPER_MULTIPLIER * self.time_halfbandwidth_product / self.time_window_duration

    @property
    def nyquist_frequency(self):
        return self.sampling_frequency / 2

    # ---- device path ---------------------------------------------------------------------
    def check_device_path(self):
        """What device_spectra checks before it computes anything (Connectivity.from_multitaper defers the transform)."""
        from . import _lib
        _lib.require_gpu()
        if self.detrend_type not in _lib.DETREND:
            raise ValueError(f"Invalid trend type '{self.detrend_type}' is not supported.\n"
                             "Valid options are 'linear'/'l', 'constant'/'c' or None.")

    def settle_device_checks(self, precision):
        """The read-backs a transform with ``defer_checks=True`` left pending -- the NaN / infinity flag of the constructor's scan
        (raised here as the constructor's warning) and the quality statistic of the planes format -- in ONE small device-to-host copy,
        at a moment the caller synchronises anyway (the first download of a result).  Returns True when the spectra of ``precision``
        stand; False when the planes format failed its check: the transform has then run again into complex64 (the cached spectra
        are the new ones) and whatever the caller computed from the old ones must be computed again."""
        pend = (self._deferred_checks or {}).pop(precision, None)
        if pend is None:
            return True
        import torch
        from . import _lib
        flag, quality, l2_min, redo = pend
        vals = torch.stack([t.reshape(()).to(torch.float32) for t in (flag, quality) if t is not None]).cpu().tolist()
        if flag is not None and vals[0] != 0.0:
            warnings.warn(_NONFINITE_WARNING, UserWarning, stacklevel=4)
        if quality is not None:
            typical = vals[-1] * l2_min
            if not typical >= _lib.PLANES_MIN_TYPICAL:
                self.device_format_note = (
                    f"the typical coefficient of a channel would be {typical:.3g} in the scaled units of the two-piece f16 "
                    f"format (limit {_lib.PLANES_MIN_TYPICAL:g}: a sample far outside the channel's usual range): "
                    "spectra kept as complex64")
                logger.warning("spectral_connectivity_amd: " + self.device_format_note)
                self._device_spectra[precision] = redo()
                return False
        return True

    def device_spectra(self, device=None, precision=None, planes_hint=None, defer_checks=False):
        """Run stage A on the GPU; returns (and caches, per precision) the HBM-resident one-sided spectra.

        ``defer_checks`` (float32 engine): the two 4-byte read-backs of a fresh object -- the NaN / infinity flag of the deferred
        constructor scan and the planes format's quality statistic -- are not waited for here (each idled the GPU long enough for
        the clock to fall back before stage B); the caller settles them with settle_device_checks() at its first download and
        computes again in the rare case the format is withdrawn (Connectivity._measure does).

        ``precision``: "float32" -- the fused f32 transform of the headline path (complex64 spectra) -- or "float64" --
        the reference's own arithmetic (float64 windows, tapers and FFT; complex128 spectra).  None: what
        ``options.precision`` gives for a call without a dtype, i.e. float64 like the reference unless forced.
        ``planes_hint`` (float32 only): the accumulator families the caller is about to request; see
        engine.multitaper_spectra (the spectra may then be held as f16 pieces and decoded to complex64 on demand)."""
        from . import options
        if precision is None:
            precision = options.engine_precision(None)
        if self._device_spectra is None:
            self._device_spectra = {}
        from . import _hosts
        if precision not in self._device_spectra and _hosts.kind() == "numpy":
            from . import numpy_api                      # the torch-free host (SC_HIP_HOST=numpy): same library, NumPy buffers
            self.check_device_path()
            self._device_spectra[precision] = numpy_api.multitaper_spectra(self, precision, planes_hint)
        if precision not in self._device_spectra:
            import torch
            from . import _lib, engine
            _lib.require_gpu()
            if self.detrend_type not in _lib.DETREND:
                raise ValueError(f"Invalid trend type '{self.detrend_type}' is not supported.\n"
                                 "Valid options are 'linear'/'l', 'constant'/'c' or None.")
            if np.iscomplexobj(self.time_series):
                # the reference's generic fft takes complex series (transforms.py:1402-1405): see _complex_device_spectra
                self._device_spectra[precision] = self._complex_device_spectra(device, precision)
                return self._device_spectra[precision]
            dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
            tapers = np.asarray(self.tapers, dtype=np.float64)             # (L, K), * sqrt(fs)
            logger.info(self)
            pending_flag = [None]

            def device_scan(t):
                # the constructor's NaN / infinity scan, deferred to the uploaded copy (see __init__)
                if self._finite_checked:
                    return
                self._finite_checked = True
                flag = torch.zeros((1,), dtype=torch.int32, device=dev)
                fn = _lib.load().sc_nonfinite_f64 if t.dtype == torch.float64 else _lib.load().sc_nonfinite_f32
                _lib.check(fn(t.data_ptr(), t.numel(), flag.data_ptr(), torch.cuda.current_stream().cuda_stream),
                           "sc_nonfinite")
                if defer_checks and precision != "float64":
                    pending_flag[0] = flag                 # read with the first download (settle_device_checks)
                elif int(flag.item()):
                    warnings.warn(_NONFINITE_WARNING, UserWarning, stacklevel=4)

            on_device = self.time_series.tensor if isinstance(self.time_series, _DeviceSeries) else None
            if on_device is not None:
                dev = on_device.device
            if precision == "float64":
                x = (on_device.to(torch.float64).contiguous() if on_device is not None
                     else torch.from_numpy(np.ascontiguousarray(self.time_series, dtype=np.float64)).to(dev))
                device_scan(x)
                h = torch.from_numpy(np.ascontiguousarray(tapers.T / self.sampling_frequency)).to(dev)
                self._device_spectra[precision] = engine.multitaper_spectra_f64(
                    x, h, self.n_time_samples_per_window, self.n_time_samples_per_step,
                    self.n_fft_samples, self.n_time_windows, self.detrend_type)
            else:
                ts = self.time_series
                n_signals = ts.shape[2]
                # (an odd channel count rides on one zero pad channel -- up to the planes format's 1024 signals since round 6, 256 before)
                n_alloc = n_signals + 1 if (n_signals % 2 and n_signals + 1 <= _lib.PLANES_FORMAT_MAX_CHANNELS) else n_signals
                if on_device is not None and ts.dtype == np.float32:
                    # already in HBM, float32: used in place (an odd channel count gets its zero pad channel in engine.multitaper_spectra)
                    x = on_device.contiguous()
                    device_scan(x)
                    n_signals = None if n_alloc != ts.shape[2] else n_signals
                elif ts.dtype == np.float64 and ts.size:
                    # float64 input: uploaded as it is and converted on the device (sc_timeseries_to_f32), which also takes a
                    # per-(trial, signal) constant out in float64 BEFORE the cast when a detrend is active -- every window's
                    # own detrend removes any constant, and a DC offset 1e5 times the signal (raw EEG / MEG) would otherwise
                    # cost the float32 copy all but two digits of the signal -- and appends the zero pad channel of odd counts
                    xd = on_device.contiguous() if on_device is not None else torch.from_numpy(np.ascontiguousarray(ts)).to(dev)
                    device_scan(xd)
                    x = torch.empty(ts.shape[:2] + (n_alloc,), dtype=torch.float32, device=dev)
                    _lib.check(_lib.load().sc_timeseries_to_f32(xd.data_ptr(), ts.shape[0], ts.shape[1], n_signals,
                                                                int(self.detrend_type is not None), x.data_ptr(), n_alloc,
                                                                torch.cuda.current_stream().cuda_stream), "sc_timeseries_to_f32")
                    del xd
                else:
                    x_host = np.ascontiguousarray(np.asarray(ts), dtype=np.float32)
                    if n_alloc != n_signals:
                        # odd channel count: ONE all-zero channel is appended on the host, before the upload, so that the
                        # rows of the spectra stay 16-byte aligned for the one-pass stage-B kernels (engine.DeviceSpectra)
                        x_host = np.concatenate([x_host, np.zeros(x_host.shape[:2] + (1,), dtype=np.float32)], axis=2)
                    x = torch.from_numpy(x_host).to(dev)
                    device_scan(x)
                h = torch.from_numpy(np.ascontiguousarray(tapers.T / self.sampling_frequency, dtype=np.float32)).to(dev)
                sp = engine.multitaper_spectra(
                    x, h, self.n_time_samples_per_window, self.n_time_samples_per_step,
                    self.n_fft_samples, self.n_time_windows, self.detrend_type, n_signals=n_signals,
                    planes_hint=planes_hint)
                self.device_format_note = None

                def redo_complex64():
                    return engine.multitaper_spectra(
                        x, h, self.n_time_samples_per_window, self.n_time_samples_per_step,
                        self.n_fft_samples, self.n_time_windows, self.detrend_type, n_signals=n_signals, planes_hint=None)
                if defer_checks and (pending_flag[0] is not None or (sp.P is not None and sp.quality is not None)):
                    if self._deferred_checks is None:
                        self._deferred_checks = {}
                    checked = sp.P is not None and sp.quality is not None
                    self._deferred_checks[precision] = (pending_flag[0], sp.quality if checked else None,
                                                        sp.taper_l2_min if checked else None, redo_complex64)
                elif sp.P is not None and sp.quality is not None:
                    # The planes format takes ONE scale per channel from the range of its samples: a channel with an artefact
                    # hundreds of times its typical amplitude would hold the quiet windows' coefficients near the f16
                    # subnormals.  The scale pass measured the typical magnitude on the way; below the limit the transform
                    # runs again into complex64 (one small read-back: the first transform of an object, never a step of a loop).
                    typical = sp.planes_typical_coefficient()
                    if not typical >= _lib.PLANES_MIN_TYPICAL:
                        self.device_format_note = (
                            f"the typical coefficient of a channel would be {typical:.3g} in the scaled units of the two-piece f16 "
                            f"format (limit {_lib.PLANES_MIN_TYPICAL:g}: a sample far outside the channel's usual range): "
                            "spectra kept as complex64")
                        logger.warning("spectral_connectivity_amd: " + self.device_format_note)
                        del sp
                        sp = redo_complex64()
                self._device_spectra[precision] = sp
        if not defer_checks and self._deferred_checks:
            self.settle_device_checks(precision)          # (a caller that cannot compute again: settled before it sees the spectra)
        return self._device_spectra[precision]

    def _complex_device_spectra(self, device, precision):
        """Complex-valued series z = a + i b.  Window extraction, detrend (real regressors: real and imaginary part are
        fitted separately, transforms.py:1903-1909), taper and FFT are all linear, so Z = FFT(a) + i FFT(b): the device
        transforms the 2 C real series (a | b) with the real-input kernels and a pointwise pass assembles the two-sided
        spectrum -- bins 0 .. N/2 as A + i B, the others from the conjugate mirrors of A and B -- into a DeviceSpectra with
        ``real_input=False`` (all N bins stored, like uploaded coefficients)."""
        import torch
        from . import engine
        ts = np.asarray(self.time_series)
        C = ts.shape[2]
        with warnings.catch_warnings():
            # (the checks of the 2 C real parts would repeat this object's warnings -- non-finite samples -- or raise ones that
            #  are not true of the complex series: "may be transposed" for T < 2 C)
            warnings.simplefilter("ignore", UserWarning)
            parts = Multitaper(np.concatenate([ts.real, ts.imag], axis=2), sampling_frequency=self.sampling_frequency,
                               time_halfbandwidth_product=self.time_halfbandwidth_product, detrend_type=self.detrend_type,
                               start_time=self.start_time, n_fft_samples=self._n_fft_samples, tapers=self._tapers,
                               n_tapers=self._n_tapers, n_time_samples_per_window=self.n_time_samples_per_window,
                               n_time_samples_per_step=self.n_time_samples_per_step, is_low_bias=self.is_low_bias)
        parts._finite_checked = self._finite_checked
        sp2 = parts.device_spectra(device, precision)
        self._finite_checked = True
        X2 = sp2.coefficients()                                   # (F, W, R, K, 2 C) one-sided
        N, F = sp2.n_fft, sp2.F
        A, B = X2[..., :C], X2[..., C:]
        mirror = torch.arange(N - F, 0, -1, device=X2.device)      # bin f = F .. N - 1 takes the conjugate of bin N - f
        C_alloc = C if (sp2.f64 or C % 2 == 0 or C + 1 > 256) else C + 1
        X = torch.zeros((N,) + tuple(X2.shape[1:4]) + (C_alloc,), dtype=X2.dtype, device=X2.device)
        X[:F, ..., :C] = A + 1j * B
        if N > F:
            X[F:, ..., :C] = torch.conj(A[mirror]) + 1j * torch.conj(B[mirror])
        W, R, K = X.shape[1:4]
        return engine.DeviceSpectra(X, (N, W, R, K, C), (W * R * K * C_alloc, R * K * C_alloc, K * C_alloc, C_alloc), N,
                                    real_input=False, C_alloc=C_alloc)

    def fft(self):
        """Fourier coefficients (n_time_windows, n_trials, n_tapers, n_fft_samples, n_signals).

        Drop-in for reference transforms.py:1147-1171: complex128, two-sided, computed in float64 on the device
        (float32 if ``options.precision == "float32"``) as a one-sided real transform; the negative-frequency half is
        the conjugate mirror (real input) and is filled on the host only for this export.
        """
        from . import _hosts
        if _hosts.kind() == "numpy":
            from . import numpy_api
            return numpy_api.fft(self)
        sp = self.device_spectra()
        one = sp.coefficients().cpu().numpy().astype(np.complex128, copy=False)          # (F, W, R, K, C)
        one = np.moveaxis(one, 0, 3)                            # (W, R, K, F, C)
        if not sp.real_input:                                   # complex series: all N bins are stored
            return np.ascontiguousarray(one)
        N = self.n_fft_samples
        out = np.empty(one.shape[:3] + (N, one.shape[-1]), dtype=np.complex128)
        F = one.shape[3]
        out[..., :F, :] = one
        if N > F:
            out[..., F:, :] = np.conj(one[..., N - F:0:-1, :])
        return out
