"""Labelled (xarray) front end: time series in, DataArray / Dataset of connectivity measures out.

Mirror of the reference's ``wrapper`` module (reference wrapper.py:17-287): same function names,
arguments, dimension names (``time``, ``frequency``, ``source``, ``target``), ``mt_*`` attributes and
error behaviour.  Differences: every requested measure is computed from ONE ``Connectivity`` object, so
the spectra are transformed once and measures that share accumulator planes share one device pass (the
reference builds a new ``Connectivity`` -- and re-reduces the coefficients -- per method); and ``xarray``
is optional: when it cannot be imported the results are the minimal labelled arrays of ``_labelled.py`` (same
``values`` / ``dims`` / ``coords`` / ``attrs`` / ``name``).
"""
import inspect
from logging import getLogger

import numpy as np

from . import _hosts
from .connectivity import Connectivity as _BaseConnectivity
from .transforms import Multitaper


class _ConnectivityOfTheHost:
    """``Connectivity`` of the host this process uses (SC_HIP_HOST, _hosts.py): the torch-free class needs no torch import."""

    def __getattr__(self, name):
        if _hosts.kind() == "numpy":
            from .numpy_api import Connectivity as cls
        else:
            cls = _BaseConnectivity
        return getattr(cls, name)


Connectivity = _ConnectivityOfTheHost()

logger = getLogger(__name__)

# not measures, or not expressible as (time, frequency, source, target) -- reference wrapper.py:225-246
_NOT_IN_DATASET = {
    "delay", "n_observations", "frequencies", "all_frequencies", "global_coherence", "from_multitaper",
    "phase_slope_index", "subset_pairwise_spectral_granger_prediction", "group_delay", "canonical_coherence",
    "directed_transfer_function", "directed_coherence", "partial_directed_coherence",
    "generalized_partial_directed_coherence", "direct_directed_transfer_function",
    "blockwise_spectral_granger_prediction",
}
_MT_SKIP = {"time_series", "fft", "tapers", "frequencies", "time"}


def _xarray():
    """``xarray`` when it can be imported (the reference's dependency), else the vendored minimal labelled arrays of
    ``_labelled.py`` -- same attribute names (values / dims / coords / attrs / name), ``sel`` / ``isel`` / ``squeeze``."""
    try:
        import xarray
        return xarray
    except ImportError:
        from . import _labelled
        return _labelled


def _check_method(method):
    if method in ("group_delay", "canonical_coherence") or "directed" in method:
        raise ValueError(
            f"The method '{method}' is not supported by the xarray interface. "
            f"Please use the Connectivity class directly instead:\n\n"
            f"from spectral_connectivity_amd import Connectivity  # in place of: from spectral_connectivity import Connectivity\n"
            f"conn = Connectivity.from_multitaper(m)\n"
            f"result = conn.{method}()\n")


def _to_dataarray(xr, m, connectivity, method, signal_names, squeeze, **kwargs):
    n_signals = m.time_series.shape[-1]
    names = list(np.arange(n_signals).astype(str)) if signal_names is None else signal_names
    values = getattr(connectivity, method)(**kwargs)
    if n_signals > 2 and squeeze:
        logger.warning(f"Squeeze is on, but there are {n_signals} pairs!")
    if method == "power":
        out = xr.DataArray(values, coords=[connectivity.time, connectivity.frequencies, names],
                           dims=["time", "frequency", "source"])
    elif n_signals == 2 and squeeze:
        out = xr.DataArray(values[..., 0, -1], coords=[connectivity.time, connectivity.frequencies],
                           dims=["time", "frequency"])
    else:
        out = xr.DataArray(values, coords=[connectivity.time, connectivity.frequencies, names, names],
                           dims=["time", "frequency", "source", "target"])
    out.name = method
    for attr in dir(m):
        if attr.startswith("_") or attr in _MT_SKIP:
            continue
        out.attrs["mt_" + attr] = getattr(m, attr)      # the prefix keeps xarray's .dt accessor out of the way
    return out


def connectivity_to_xarray(m, method="coherence_magnitude", signal_names=None, squeeze=False, **kwargs):
    """One connectivity measure of a ``Multitaper`` as a labelled ``xarray.DataArray``
    (dims time, frequency, source[, target]; ``squeeze`` drops the signal axes of a two-signal measure)."""
    _check_method(method)
    return _to_dataarray(_xarray(), m, Connectivity.from_multitaper(m), method, signal_names, squeeze, **kwargs)


def multitaper_connectivity(time_series, sampling_frequency, time_window_duration=None, method=None,
                            signal_names=None, squeeze=False, connectivity_kwargs=None, **kwargs):
    """Multitaper transform + connectivity measures in one call.

    ``method``: one name -> ``DataArray``; a list, or None for every measure the labelled interface can
    express -> ``Dataset`` with one variable per measure.  Other keyword arguments go to ``Multitaper``,
    ``connectivity_kwargs`` to the measure.
    """
    connectivity_kwargs = connectivity_kwargs or {}
    single = isinstance(method, str)
    if method is None:
        methods = [name for name, _ in inspect.getmembers(_BaseConnectivity, predicate=inspect.isfunction)
                   if not name.startswith("_") and name not in _NOT_IN_DATASET]
    else:
        methods = [method] if single else list(method)
    if len(methods) == 1:
        _check_method(methods[0])
    xr = _xarray()
    # `dtype` (an extension: the reference's wrapper always builds its Connectivity with the default complex128) picks the
    # engine like Connectivity.from_multitaper(dtype=...): numpy.complex64 = the float32 engine
    dtype = kwargs.pop("dtype", None)
    m = Multitaper(time_series=time_series, sampling_frequency=sampling_frequency,
                   time_window_duration=time_window_duration, **kwargs)
    # shared: one transform, shared accumulator passes
    connectivity = Connectivity.from_multitaper(m) if dtype is None else Connectivity.from_multitaper(m, dtype=dtype)
    if len(methods) > 1:
        connectivity._prepare(methods)
    out = xr.Dataset()
    for name in methods:
        try:
            _check_method(name)
            out[name] = _to_dataarray(xr, m, connectivity, name, signal_names, squeeze, **connectivity_kwargs)
        except NotImplementedError as exc:
            if len(methods) == 1:
                raise exc
            logger.warning(f"{name} is not implemented in xarray")
    if single and methods[0] in out:
        return out[methods[0]]
    return out
