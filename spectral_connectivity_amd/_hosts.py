"""Which host drives libsc_hip.so in this process: ``SC_HIP_HOST=torch`` (engine.py: PyTorch owns buffers, streams, collectives) or
``SC_HIP_HOST=numpy`` (numpy_host.py / numpy_api.py: ctypes + NumPy, the reference's own dependencies and nothing else).  Unset:
torch when it is importable, NumPy otherwise.  One process uses one host (two HIP runtimes cannot share it: _lib.load())."""
import importlib.util
import os

_kind = None


def kind():
    global _kind
    if _kind is None:
        want = os.environ.get("SC_HIP_HOST", "").strip().lower()
        if want not in ("", "torch", "numpy"):
            raise ValueError(f"SC_HIP_HOST={want!r}: expected 'torch' or 'numpy'")
        if not want:
            want = "torch" if importlib.util.find_spec("torch") is not None else "numpy"
        _kind = want
    return _kind
