"""Package-wide switches.  Plain module attributes: set them before the objects they affect are built.

precision
    Which device engine ``Multitaper`` / ``Connectivity`` run on.
    ``"dtype"`` (default): the ``dtype`` argument of ``Connectivity`` / ``Connectivity.from_multitaper`` decides, as it
    decides the arithmetic of the reference's cross-spectral products (connectivity.py:277-285, :1799-1822) --
    ``numpy.complex128`` (the reference's default) runs the float64 engine (float64 transform, fp64 matrix-core
    cross-spectra, float64 measures: the reference's own arithmetic), ``numpy.complex64`` the float32 engine (the
    headline path: fused f32 transform, bf16x3 / f32 matrix-core cross-spectra, fp64 epilogue).  A bare
    ``Multitaper.fft()`` is float64 like the reference's.
    ``"float32"`` / ``"float64"``: force one engine whatever ``dtype`` says.
    Environment: ``SC_HIP_PRECISION`` sets the initial value.

one_sample_fisher_z
    ``"reference"`` (default): ``statistics.coherence_fisher_z_transform(c, n)`` evaluates ``coherence_bias(0) = -1/2``
    for the absent second sample exactly like the reference (statistics.py:147-203), so one-sample z-scores are NaN,
    ``Connectivity.group_delay()`` is NaN for every pair and ``Connectivity.delay()`` returns the constants 2 pi k --
    the reference's outputs, pinned by tests/golden/f11_post.npz.
    ``"unbiased"``: the absent sample has no bias; group_delay / delay then report the delays they describe.

finite_check
    Where the constructor's NaN / infinity scan of the time series (reference transforms.py:746-753) runs.
    ``"device"`` (default): series of at least ``FINITE_CHECK_DEVICE_MIN`` samples are scanned on the device next to
    their upload (one read at HBM rate instead of 18 ms of one core at the cfg3 shape) and the reference's warning is
    raised by the first transform; smaller series are scanned by the constructor like the reference's.
    ``"host"``: always in the constructor.  Environment: ``SC_HIP_FINITE_CHECK``.

anticipate_phase_lag
    ``True`` (default): when the first request of a float32-engine ``Connectivity`` is a cross-spectral measure (power,
    coherency, coherence, ...) on a shape whose spectra are held in the planes format (44 ... 256 signals, >= 256 MB of
    spectra: _lib.planes_format_applies), the same pass over the spectra also sums the per-observation |Im s| plane that
    ``weighted_phase_lag_index`` needs -- the matrix-core kernel produces both in one launch (sc_fused2.hip) -- so a wPLI that
    follows costs an epilogue (0.1 ms at the BASELINE shape) instead of a second pass over 6.5 GB of spectra (3.5 ms).
    What it costs a caller who never asks for a phase-lag measure: the |Im s| role of that launch (2.4 -> 3.5 ms of stage
    B at 128 signals x 7000 observations x 903 bins).  ``False``: every request accumulates exactly the families it needs
    (round-4 behaviour).  Environment: ``SC_HIP_ANTICIPATE=0``.
"""
import os

precision = os.environ.get("SC_HIP_PRECISION", "dtype")
one_sample_fisher_z = "reference"
finite_check = os.environ.get("SC_HIP_FINITE_CHECK", "device")
FINITE_CHECK_DEVICE_MIN = 1 << 22
anticipate_phase_lag = os.environ.get("SC_HIP_ANTICIPATE", "1") != "0"


def engine_precision(dtype=None):
    """'float32' or 'float64' for a Connectivity ``dtype`` (None: no dtype in play -> the reference's float64)."""
    import numpy as np
    if precision in ("float32", "float64"):
        return precision
    if precision != "dtype":
        raise ValueError(f"options.precision must be 'dtype', 'float32' or 'float64', got {precision!r}")
    if dtype is None:
        return "float64"
    dt = np.dtype(dtype)
    if dt == np.complex64:
        return "float32"
    if dt == np.complex128:
        return "float64"
    raise ValueError(f"dtype must be numpy.complex64 or numpy.complex128, got {dt}")
