"""CPU-only checks of the drop-in boundary: libsc_hip.so loads and exports every symbol that
include/sc_hip.h declares; argument validation works without a GPU (no compute calls)."""
import os
import re
from ctypes import byref, c_int64

import pytest

from spectral_connectivity_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "sc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sc_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 12
    for name in names:
        assert hasattr(lib, name), f"libsc_hip.so does not export {name}"
    assert sorted(_lib.SYMBOLS) == sorted(n for n in names), "ctypes table out of sync with sc_hip.h"
    assert lib.sc_abi_version() == _lib.SC_ABI_VERSION


def test_loaded_from_tree():
    assert os.path.dirname(_lib.library_path()) == os.path.join(ROOT, "spectral_connectivity_amd")


def test_accum_layout_and_validation():
    lib = _lib.load()
    d = _lib.SpectraDesc(n_freq=129, n_windows=7, n_trials=10, n_tapers=7, n_signals=128,
                         stride_freq=7 * 10 * 7 * 128, stride_window=10 * 7 * 128, stride_trial=7 * 128,
                         stride_taper=128, reduce_window=0, reduce_trial=1, reduce_taper=1, reserved=0)
    n_bins, fpb, n_groups, n_obs = c_int64(), c_int64(), c_int64(), c_int64()
    planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
    assert lib.sc_accum_layout(byref(d), planes, byref(n_bins), byref(fpb), byref(n_groups), byref(n_obs)) == 0
    assert (n_bins.value, n_groups.value, n_obs.value) == (7 * 129, 7, 70)
    assert fpb.value == 3 * 36 * 256          # 3 planes x 36 upper 16x16 tiles
    # NULL pointers are rejected with SC_EINVAL and a message, before touching the device
    rc = lib.sc_taper_windows_f32(None, 8, 1, 1, 8, 8, 1, 8, None, 1, 1, None, None)
    assert rc == -1 and b"NULL" in lib.sc_last_error()
    rc = lib.sc_measure_f32(None, 1, 1, 1, 1, 0, None, None)
    assert rc == -1
    with pytest.raises(_lib.HipEngineError):
        _lib.check(rc, "sc_measure_f32")


def test_no_cpu_fallback_without_gpu():
    import numpy as np
    import torch

    from spectral_connectivity_amd import Multitaper
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = Multitaper(np.zeros((64, 2, 2)), sampling_frequency=100)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.fft()


def test_gpu_switch_of_the_reference_is_honoured():
    """SPECTRAL_CONNECTIVITY_ENABLE_GPU (reference transforms.py:405-439): "true" loads the HIP engine at import and a
    library that cannot be loaded is a RuntimeError right there; any other value asks for the NumPy backend this
    package does not have and is refused at the first computation; unset means load on first use."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(code, **env):
        e = dict(os.environ, **env)
        return subprocess.run([sys.executable, "-c", code], cwd=root, env=e, capture_output=True, text=True, timeout=300)

    ok = run("import spectral_connectivity_amd, sys; from spectral_connectivity_amd import _lib; "
             "print(_lib._lib is not None)", SPECTRAL_CONNECTIVITY_ENABLE_GPU="true")
    assert ok.returncode == 0 and ok.stdout.strip().endswith("True"), ok.stderr[-2000:]
    bad = run("import spectral_connectivity_amd", SPECTRAL_CONNECTIVITY_ENABLE_GPU="true",
              SC_HIP_LIB="/nonexistent/libsc_hip.so")
    assert bad.returncode != 0 and "explicitly requested via SPECTRAL_CONNECTIVITY_ENABLE_GPU='true'" in bad.stderr
    cpu = run("import numpy as np, spectral_connectivity_amd as sc\n"
              "m = sc.Multitaper(np.zeros((64, 1, 2)))\n"
              "try:\n    m.fft()\nexcept RuntimeError as exc:\n    print('refused:', exc)\n",
              SPECTRAL_CONNECTIVITY_ENABLE_GPU="false")
    assert cpu.returncode == 0 and "refused:" in cpu.stdout and "NumPy backend" in cpu.stdout, cpu.stdout + cpu.stderr[-2000:]
