"""MI355X-native multitaper spectral connectivity.

Drop-in for the ``Multitaper`` / ``Connectivity`` hot path of
Eden-Kramer-Lab/spectral_connectivity (reference __init__.py:34-44 export list), executed
by hand-written HIP kernels for gfx950 + rocFFT behind the C ABI of ``include/sc_hip.h``.

The public names are resolved on first use (PEP 562): ``import spectral_connectivity_amd.numpy_host`` -- the torch-free
ctypes + NumPy host of the same library -- must not drag in the PyTorch host that ``Connectivity`` sits on.
"""
import importlib

__version__ = "0.1.0"

_EXPORTS = {
    "Connectivity": ".connectivity",
    "Multitaper": ".transforms",
    "MultitaperParameters": ".transforms",
    "prepare_time_series": ".transforms",
    "suggest_parameters": ".transforms",
    "estimate_frequency_resolution": ".transforms",
    "estimate_n_tapers": ".transforms",
    "get_compute_backend": ".utils",
    "multitaper_connectivity": ".wrapper",
}
__all__ = list(_EXPORTS)

# the reference's import-time backend switch (transforms.py:405-439): SPECTRAL_CONNECTIVITY_ENABLE_GPU=true loads the
# engine NOW and fails here if it cannot be loaded; unset / anything else costs nothing at import
from . import _lib as _binding  # noqa: E402  (ctypes table only: no torch, no library load)

_binding.honour_gpu_switch()


def __getattr__(name):
    if name in _EXPORTS:
        module = _EXPORTS[name]
        if name == "Connectivity":
            # two hosts of the same C ABI (SC_HIP_HOST=torch|numpy, _hosts.py): the torch-free one needs NumPy + SciPy only
            from . import _hosts
            module = ".numpy_api" if _hosts.kind() == "numpy" else module
        value = getattr(importlib.import_module(module, __name__), name)
        globals()[name] = value
        return value
    try:                                   # submodules: spectral_connectivity_amd.transforms, .engine, ...
        return importlib.import_module("." + name, __name__)
    except ModuleNotFoundError as exc:
        if exc.name != f"{__name__}.{name}":
            raise                          # a real submodule whose own import failed (torch missing, ...): say so
        raise AttributeError(f"module {__name__!r} has no attribute {name!r}") from None


def __dir__():
    return sorted(set(globals()) | set(__all__))
