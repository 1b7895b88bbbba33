"""Backend report, same role as reference utils.py:8-167 (get_compute_backend)."""


def get_compute_backend():
    """Describe the compute backend: always the HIP engine (no CPU fallback exists)."""
    import os
    info = {"backend": "hip", "gpu_enabled": True, "gpu_switch": os.environ.get("SPECTRAL_CONNECTIVITY_ENABLE_GPU"),
            "library": None, "gpu_available": False, "device_name": None,
            "n_devices": 0, "message": ""}            # keys of the reference's report + library / n_devices
    try:
        import torch

        from . import _lib
        _lib.load()
        info["library"] = _lib.library_path()
        info["gpu_available"] = bool(torch.cuda.is_available())
        if info["gpu_available"]:
            info["n_devices"] = torch.cuda.device_count()
            info["device_name"] = torch.cuda.get_device_name(0)
            info["message"] = f"HIP engine on {info['device_name']}"
        else:
            info["message"] = "libsc_hip.so loaded but no ROCm GPU is visible; compute calls will raise"
    except Exception as exc:  # report, never hide
        info["message"] = f"HIP engine unavailable: {exc}"
    return info
