"""ctypes binding of the C ABI in include/sc_hip.h (libsc_hip.so).

The HIP library is the ONLY compute path of this package: if it cannot be loaded the import
of any compute entry point raises -- there is no NumPy/CPU fallback.

Two hosts sit on this binding.  The PyTorch host (engine.py: torch owns HBM buffers, streams and the RCCL collectives)
imports torch BEFORE the library on purpose: the PyTorch-ROCm wheel ships its own HIP runtime / rocFFT (same SONAMEs
as /opt/rocm); loading it first makes libsc_hip.so bind to that single runtime, so device pointers of torch tensors
are valid inside our kernels.  The NumPy host (numpy_host.py: memory, copies and streams through the library's own
sc_device_alloc / sc_memcpy_* / sc_stream_* calls) never imports torch; the library then binds to /opt/rocm's runtime.
One process uses one of the two: whichever loads the library first decides the runtime it is bound to.
"""
import ctypes
import os
from ctypes import POINTER, Structure, byref, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint32, c_void_p

import sys

from . import _build

c_int64_p = POINTER(c_int64)

# ---- constants mirrored from include/sc_hip.h -------------------------------------------
SC_ABI_VERSION = 6
GRANGER_KEEP_OUTPUT = 1
DETREND = {None: 0, "constant": 1, "c": 1, "linear": 2, "l": 2}
MVAR_DTF, MVAR_DC, MVAR_PDC, MVAR_GPDC, MVAR_DDTF, MVAR_TRANSFER, MVAR_COEFFICIENTS, MVAR_NOISE_COVARIANCE = range(8)
PLANE_CSM, PLANE_ABS_IM, PLANE_IM_SQ, PLANE_SIGN_IM, PLANE_UNIT = 0x01, 0x02, 0x04, 0x08, 0x10
# axes of (window, trial, taper) an expectation type averages over (reference connectivity.py:67-75)
EXPECTATION_AXES = {
    "time": (0,),
    "trials": (1,),
    "tapers": (2,),
    "time_trials": (0, 1),
    "time_tapers": (0, 2),
    "trials_tapers": (1, 2),
    "time_trials_tapers": (0, 1, 2),
}
RECORD_F64 = 0x100          # OR-ed into `planes` when the records handed to a consumer hold doubles (float64 engine)
(M_POWER, M_CSM, M_COHERENCY, M_COHERENCE_MAGNITUDE, M_COHERENCE_PHASE, M_IMAGINARY_COHERENCE,
 M_PLV, M_PLI, M_WPLI, M_DEBIASED_PLI2, M_DEBIASED_WPLI2, M_PPC, M_PLV_COMPLEX) = range(13)
COMPLEX_MEASURES = {M_CSM, M_COHERENCY, M_PLV_COMPLEX}
MEASURE_PLANES = {
    M_POWER: PLANE_CSM, M_CSM: PLANE_CSM, M_COHERENCY: PLANE_CSM, M_COHERENCE_MAGNITUDE: PLANE_CSM,
    M_COHERENCE_PHASE: PLANE_CSM, M_IMAGINARY_COHERENCE: PLANE_CSM,
    M_PLV: PLANE_UNIT, M_PLV_COMPLEX: PLANE_UNIT, M_PPC: PLANE_UNIT,
    M_PLI: PLANE_SIGN_IM, M_DEBIASED_PLI2: PLANE_SIGN_IM,
    M_WPLI: PLANE_CSM | PLANE_ABS_IM, M_DEBIASED_WPLI2: PLANE_CSM | PLANE_ABS_IM | PLANE_IM_SQ,
}


class Timing(Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("ms", c_float)]


class SpectraDesc(Structure):
    _fields_ = [("n_freq", c_int64), ("n_windows", c_int64), ("n_trials", c_int64),
                ("n_tapers", c_int64), ("n_signals", c_int64), ("stride_freq", c_int64),
                ("stride_window", c_int64), ("stride_trial", c_int64), ("stride_taper", c_int64),
                ("reduce_window", c_int32), ("reduce_trial", c_int32), ("reduce_taper", c_int32),
                ("reserved", c_int32)]


# every symbol include/sc_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "sc_abi_version": (c_int, []),
    "sc_last_error": (c_char_p, []),
    "sc_device_count": (c_int, [POINTER(c_int)]),
    "sc_taper_windows_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64,
                                     c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "sc_taper_windows_f64": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64,
                                     c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "sc_fft_plan_create_f64": (c_int, [POINTER(c_void_p), c_int64, c_int64]),
    "sc_fft_execute_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "sc_accumulate_f64": (c_int, [c_void_p, POINTER(SpectraDesc), c_uint32, c_uint32, c_void_p, c_void_p]),
    "sc_measure_f64": (c_int, [c_void_p, c_int64, c_int64, c_uint32, c_int64, c_int, c_void_p, c_void_p]),
    "sc_measure_multi_f32": (c_int, [c_void_p, c_int64, c_int64, c_uint32, c_int64, c_int, POINTER(c_int),
                                     POINTER(c_void_p), c_void_p]),
    "sc_measure_multi_f64": (c_int, [c_void_p, c_int64, c_int64, c_uint32, c_int64, c_int, POINTER(c_int),
                                     POINTER(c_void_p), c_void_p]),
    "sc_measure_multi_parts": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_uint32, c_int64, c_int,
                                       POINTER(c_int), POINTER(c_void_p), c_int, c_void_p]),
    "sc_measure_parts": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_uint32, c_int64, c_int, c_void_p, c_int, c_void_p]),
    "sc_timing_enable": (c_int, [c_int]),
    "sc_last_timing": (c_int, [POINTER(Timing), c_int, POINTER(c_int)]),
    "sc_multitaper_fft_supported": (c_int, [c_int64, c_int64]),
    "sc_fft_twiddles_f32": (c_int, [c_int64, c_void_p, c_void_p]),
    "sc_multitaper_fft_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64,
                                      c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "sc_timeseries_to_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p]),
    "sc_multitaper_fft_f64_supported": (c_int, [c_int64, c_int64]),
    "sc_multitaper_fft_f64": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64,
                                      c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "sc_fft_plan_create": (c_int, [POINTER(c_void_p), c_int64, c_int64]),
    "sc_fft_plan_work_bytes": (c_int, [c_void_p, POINTER(c_size_t)]),
    "sc_fft_execute": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "sc_fft_plan_destroy": (c_int, [c_void_p]),
    "sc_accum_layout": (c_int, [POINTER(SpectraDesc), c_uint32, c_int64_p, c_int64_p, c_int64_p, c_int64_p]),
    "sc_csm_accumulate_f32": (c_int, [c_void_p, POINTER(SpectraDesc), c_uint32, c_void_p, c_void_p]),
    "sc_nonlinear_accumulate_f32": (c_int, [c_void_p, POINTER(SpectraDesc), c_uint32, c_uint32, c_void_p, c_void_p]),
    "sc_fused_supported": (c_int, [c_int64]),
    "sc_fused_csm_absim_f32": (c_int, [c_void_p, POINTER(SpectraDesc), c_uint32, c_void_p, c_void_p]),
    "sc_fused_workspace_bytes": (c_int64, [POINTER(SpectraDesc), c_uint32]),
    "sc_fused_csm_absim_ws_f32": (c_int, [c_void_p, POINTER(SpectraDesc), c_uint32, c_void_p, c_void_p, c_int64,
                                          c_void_p]),
    "sc_unit_scratch_bytes": (c_int64, [POINTER(SpectraDesc)]),
    "sc_unit_accumulate_f32": (c_int, [c_void_p, POINTER(SpectraDesc), c_uint32, c_void_p, c_void_p, c_int64, c_void_p]),
    "sc_fused_planes_covered": (c_uint32, [POINTER(SpectraDesc), c_uint32]),
    "sc_fused_sign_ws_f32": (c_int, [c_void_p, POINTER(SpectraDesc), c_uint32, c_void_p, c_void_p, c_int64, c_void_p]),
    "sc_fused_unit_scratch_bytes": (c_int64, [POINTER(SpectraDesc)]),
    "sc_fused_unit_ws_f32": (c_int, [c_void_p, POINTER(SpectraDesc), c_uint32, c_void_p, c_void_p, c_int64, c_void_p,
                                     c_int64, c_void_p]),
    "sc_planes_row_bytes": (c_int64, [c_int64]),
    "sc_multitaper_fft_planes_supported": (c_int, [c_int64, c_int64, c_int64]),
    "sc_multitaper_fft_planes_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64,
                                             c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sc_planes_scales_from_series_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_double, c_void_p, c_void_p, c_void_p]),
    "sc_planes_scales_work_bytes": (c_int64, [c_int64, c_int64]),
    "sc_planes_scales_quality_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_double, c_void_p, c_void_p, c_int64, c_void_p,
                                             c_void_p]),
    "sc_planes_scales_from_spectra_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "sc_planes_from_spectra_f32": (c_int, [c_void_p, POINTER(SpectraDesc), c_void_p, c_void_p, c_void_p]),
    "sc_spectra_from_planes_f32": (c_int, [c_void_p, POINTER(SpectraDesc), c_void_p, c_void_p, c_void_p]),
    "sc_fused2_supported": (c_int, [POINTER(SpectraDesc), c_uint32]),
    "sc_debug_fused2_clock": (c_int, [POINTER(c_double)]),
    "sc_debug_reload_env": (c_int, []),
    "sc_debug_fft_plans": (c_int, [POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    "sc_fused2_csm_absim_parts_f32": (c_int, [c_void_p, POINTER(SpectraDesc), c_void_p, c_uint32, c_void_p, c_void_p, c_int64,
                                              POINTER(c_int), c_void_p]),
    "sc_fused2_csm_absim_f32": (c_int, [c_void_p, POINTER(SpectraDesc), c_void_p, c_uint32, c_void_p, c_void_p, c_int64,
                                        c_void_p]),
    "sc_comm_available": (c_int, []),
    "sc_comm_unique_id": (c_int, [c_void_p]),
    "sc_comm_create": (c_int, [c_void_p, c_int, c_int, POINTER(c_void_p)]),
    "sc_comm_destroy": (c_int, [c_void_p]),
    "sc_comm_size": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "sc_comm_allreduce_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "sc_comm_exchange_blocks_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "sc_comm_gather_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "sc_granger_workspace_bytes": (c_int, [c_int64, c_int64, c_int64, POINTER(c_size_t)]),
    "sc_granger_pairwise_f64": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_uint32, c_int64,
                                        c_void_p, c_int64, c_double, c_int, c_void_p, c_size_t, c_int, c_void_p,
                                        c_void_p, c_void_p, POINTER(c_int32), c_void_p]),
    "sc_wilson_factor_f64": (c_int, [c_void_p, c_int64, c_int64, c_double, c_int, c_void_p, c_size_t, c_void_p,
                                     c_void_p, c_void_p, POINTER(c_int32), c_void_p]),
    "sc_mvar_max_signals": (c_int, []),
    "sc_mvar_workspace_bytes": (c_int, [c_int64, c_int64, c_int64, POINTER(c_size_t)]),
    "sc_mvar_factor_f64": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_uint32, c_int64,
                                   c_double, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p,
                                   POINTER(c_int32), c_void_p]),
    "sc_mvar_measure_f64": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p, c_size_t,
                                    c_void_p]),
    "sc_global_coherence_max_signals": (c_int, []),
    "sc_global_coherence_f64": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_uint32, c_int64, c_int, c_int,
                                        c_void_p, c_void_p, c_void_p]),
    "sc_canonical_max_group": (c_int, []),
    "sc_canonical_coherence_f64": (c_int, [c_void_p, c_int64, c_int64, c_uint32, c_int64, c_void_p, c_void_p,
                                           c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sc_measure_f32": (c_int, [c_void_p, c_int64, c_int64, c_uint32, c_int64, c_int, c_void_p, c_void_p]),
    # host-pointer side (sc_memory.hip)
    "sc_device_alloc": (c_int, [POINTER(c_void_p), c_size_t, c_void_p]),
    "sc_device_free": (c_int, [c_void_p, c_void_p]),
    "sc_host_alloc": (c_int, [POINTER(c_void_p), c_size_t]),
    "sc_host_free": (c_int, [c_void_p]),
    "sc_host_register": (c_int, [c_void_p, c_size_t]),
    "sc_host_unregister": (c_int, [c_void_p]),
    "sc_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "sc_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "sc_memset_zero": (c_int, [c_void_p, c_size_t, c_void_p]),
    "sc_stream_create": (c_int, [POINTER(c_void_p)]),
    "sc_stream_destroy": (c_int, [c_void_p]),
    "sc_stream_synchronize": (c_int, [c_void_p]),
    "sc_nonfinite_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "sc_nonfinite_f64": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
}

_lib = None
_bound_with_torch = None        # True / False once the library is loaded: which HIP runtime it is bound to


class HipEngineError(RuntimeError):
    """A libsc_hip.so entry point returned a negative status."""


def library_path():
    return os.environ.get("SC_HIP_LIB", _build.LIB)


def load(torch_host=None):
    """Load (building in-tree if the sources are newer) and type libsc_hip.so.  Raises if impossible.
    ``torch_host``: the caller hands torch tensors' device pointers to the library, so torch's HIP runtime must be the
    one the library binds to (torch imported first); numpy_host passes False and never imports torch.  None: whatever the host of
    this process is (SC_HIP_HOST, _hosts.py)."""
    global _lib, _bound_with_torch
    if torch_host is None:
        from . import _hosts
        torch_host = _hosts.kind() != "numpy"
    if _lib is not None:
        if torch_host and not _bound_with_torch:
            raise RuntimeError(
                "libsc_hip.so was loaded by the NumPy host (spectral_connectivity_amd.numpy_host) and is bound to "
                "/opt/rocm's HIP runtime; the PyTorch host cannot share it in the same process (torch ships its own "
                "runtime). Import spectral_connectivity_amd (or torch) before numpy_host, or use separate processes.")
        return _lib
    if torch_host or "torch" in sys.modules:
        import torch  # noqa: F401  (must precede CDLL, see module docstring)
    path = library_path()
    if path == _build.LIB and _build.is_stale():
        try:
            _build.build(verbose=False)
        except Exception as exc:  # no hipcc, compile error: fail loudly, no CPU fallback
            if not os.path.exists(path):
                raise RuntimeError(
                    "spectral_connectivity_amd needs its HIP extension libsc_hip.so and could not "
                    f"build it ({exc}). Run `python -m spectral_connectivity_amd._build` on a machine "
                    "with ROCm's hipcc; there is no CPU fallback.") from exc
    try:
        lib = ctypes.CDLL(path)
    except OSError as exc:
        raise RuntimeError(f"cannot load HIP extension {path}: {exc}. There is no CPU fallback.") from exc
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.sc_abi_version() != SC_ABI_VERSION:
        raise RuntimeError(f"{path}: ABI version {lib.sc_abi_version()} != {SC_ABI_VERSION}")
    _lib = lib
    _bound_with_torch = "torch" in sys.modules
    return lib


def _handle():
    """The loaded library, whichever host loaded it (the PyTorch host's load() on first use otherwise)."""
    return _lib if _lib is not None else load()


def check(status, what=""):
    if status != 0:
        msg = _handle().sc_last_error()
        raise HipEngineError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")


def reload_debug_env():
    """The library reads its diagnostic switches (SC_FUSED_DEBUG, SC_FUSED_SPLIT, SC_WILSON_FFT, ...) from the environment once,
    when it is loaded; tools and tests that change one afterwards call this (no-op while the library is not loaded)."""
    if _lib is not None:
        _lib.sc_debug_reload_env()


def fft_plan_counts():
    """(rocfft_plan_create calls of the process, plans the library's pools hold, idle real-to-complex row plans among them): rocFFT
    plans are pooled by geometry and never destroyed (sc_fft_plan_destroy), so the first two are always equal and grow with the
    DISTINCT geometries only."""
    a, b, c = c_int64(0), c_int64(0), c_int64(0)
    check(_handle().sc_debug_fft_plans(byref(a), byref(b), byref(c)), "sc_debug_fft_plans")
    return a.value, b.value, c.value


def set_debug_env(name, value):
    """os.environ[name] = value (None: unset) and have the library see it."""
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = str(value)
    reload_debug_env()


def device_count():
    n = c_int(0)
    check(_handle().sc_device_count(byref(n)), "sc_device_count")
    return n.value


ENABLE_GPU_ENV = "SPECTRAL_CONNECTIVITY_ENABLE_GPU"


def gpu_switch():
    """The reference's import-time backend switch (transforms.py:405-439, connectivity.py:31-65,
    minimum_phase_decomposition.py:14-26), read the way the reference reads it: the string "true" asks for the GPU
    backend, anything else for NumPy.  Returns True ("true": the HIP engine was asked for by name), None (unset: this
    package only has the HIP engine, nothing to choose) or False (set to something else: the caller asked for the
    CPU path, which this package does not have)."""
    v = os.environ.get(ENABLE_GPU_ENV)
    return None if v is None else v == "true"


def honour_gpu_switch():
    """Import-time half of the switch (called from transforms.py like the reference's module-level block): with
    SPECTRAL_CONNECTIVITY_ENABLE_GPU=true the engine must be there NOW -- a missing / unbuildable libsc_hip.so raises
    RuntimeError at import, like the reference's missing CuPy does (transforms.py:429-434)."""
    if gpu_switch():
        try:
            load()
        except Exception as exc:
            raise RuntimeError(
                f"GPU support was explicitly requested via {ENABLE_GPU_ENV}='true', but the HIP engine "
                f"libsc_hip.so could not be loaded ({exc}). Build it with "
                "'python -m spectral_connectivity_amd._build' on a machine with ROCm's hipcc.") from exc


def require_gpu():
    """The product path runs on an MI355X only: fail loudly when none is visible."""
    if gpu_switch() is False:
        raise RuntimeError(
            f"{ENABLE_GPU_ENV}={os.environ.get(ENABLE_GPU_ENV)!r} selects the reference's NumPy backend, which "
            "spectral_connectivity_amd does not have: every computation here runs on the HIP engine. Unset the "
            "variable or set it to 'true' (or use the reference package for a CPU run).")
    from . import _hosts
    if _hosts.kind() == "numpy":
        # the torch-free host: the library's own device count (the check numpy_host.NumpyHost makes)
        n = c_int(0)
        load(torch_host=False).sc_device_count(byref(n))
        if n.value < 1:
            raise RuntimeError("spectral_connectivity_amd: no ROCm GPU is visible (sc_device_count() = 0). "
                               "This engine has no CPU fallback; run on an MI355X host.")
        return
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError(
            "spectral_connectivity_amd: no ROCm GPU is visible (torch.cuda.is_available() is False). "
            "This engine has no CPU fallback; run on an MI355X host.")
    load()


def timing_enable(on=True):
    """Library-side stage timers (hipEvents on the launch stream, sc_timing.hip)."""
    check(_handle().sc_timing_enable(int(bool(on))), "sc_timing_enable")


def last_timing(max_entries=4096):
    """[(entry point, milliseconds)] of the calls since the previous read, in call order (waits for them)."""
    buf = (Timing * max_entries)()
    n = c_int(0)
    check(_handle().sc_last_timing(buf, max_entries, byref(n)), "sc_last_timing")
    return [(buf[i].name.decode(), float(buf[i].ms)) for i in range(n.value)]


# ---- planes format (sc_fused2.hip): when stage A writes it -- shared by the PyTorch host (engine.py) and the NumPy host ----------
PLANES_FORMAT_FAMILIES = (PLANE_CSM, PLANE_CSM | PLANE_ABS_IM, PLANE_CSM | PLANE_ABS_IM | PLANE_IM_SQ,
                          PLANE_SIGN_IM)   # what sc_fused2.hip accumulates


PLANES_FORMAT_MIN_CHANNELS = 44
PLANES_FORMAT_MAX_CHANNELS = 1024      # F2_MAX_SIGNALS of csrc/sc_fused2.hip
PLANES_MIN_TYPICAL = 2.5       # SC_PLANES_MIN_TYPICAL of include/sc_hip.h: smallest typical coefficient (scaled units) the format is used for


def planes_format_applies(n_window, n_fft, n_alloc, planes_hint, spectra_bytes=None):
    """Does stage A write the planes format for a caller that will ask for the accumulator families ``planes_hint``?
    ``spectra_bytes``: size of the complex64 spectra of the request, when the caller knows it.

    Round 5: for every family sc_fused2.hip accumulates the answer depends on the SHAPE of the request alone -- not on which of
    those families is asked for first -- so that a ``Connectivity`` takes the same kernels whatever the order of the calls (the
    reference computes every measure from the same coefficients, connectivity.py:463-526).  The format is taken from 44 signals on
    (where CSM + |Im s|, the BASELINE pair coherence + wPLI, first wins: profiles/r04_shape_sweep.txt; at 44 ... 60 signals the
    (Im s)^2 pass alone would be 0.7 ms faster on complex64, at 130 signals CSM alone 0.7 ms -- the price of one path) for requests
    of at least 256 MB of spectra (a few tens of MB are three short launches either way and the scale pass would only add two).
    Families outside the format (the unit-phasor plane of PLV / PPC) and callers without a hint (Granger, canonical and global
    coherence: CSM of every bin, window lengths beyond the format) keep complex64."""
    if planes_hint not in PLANES_FORMAT_FAMILIES or os.environ.get("SC_PLANES_FORMAT", "1") == "0":
        return False
    lo = PLANES_FORMAT_MIN_CHANNELS
    forced = os.environ.get("SC_PLANES_MIN_CHANNELS")               # (tests: the format from this many channels on, whatever the size)
    if forced is not None:
        lo = int(forced)
    elif spectra_bytes is not None and spectra_bytes < (256 << 20):
        return False
    # (up to 1024 signals since round 6: sc_fused2.hip plans its launches over any number of 32-channel blocks; 256 before)
    return lo <= n_alloc <= PLANES_FORMAT_MAX_CHANNELS and bool(_handle().sc_multitaper_fft_planes_supported(n_window, n_fft, n_alloc))
