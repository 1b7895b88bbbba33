"""MI355X-native multitaper spectral connectivity.

Drop-in for the ``Multitaper`` / ``Connectivity`` hot path of
Eden-Kramer-Lab/spectral_connectivity (reference __init__.py:34-44 export list), executed
by hand-written HIP kernels for gfx950 + rocFFT behind the C ABI of ``include/sc_hip.h``.
"""
from .connectivity import Connectivity
from .transforms import (
    Multitaper,
    MultitaperParameters,
    estimate_frequency_resolution,
    estimate_n_tapers,
    prepare_time_series,
    suggest_parameters,
)
from .utils import get_compute_backend
from .wrapper import multitaper_connectivity

__version__ = "0.1.0"

__all__ = [
    "Connectivity",
    "Multitaper",
    "MultitaperParameters",
    "prepare_time_series",
    "suggest_parameters",
    "estimate_frequency_resolution",
    "estimate_n_tapers",
    "get_compute_backend",
    "multitaper_connectivity",
]
