"""The golden suite of tests/test_gpu_parity.py a FOURTH time: on the torch-free host (``SC_HIP_HOST=numpy``: numpy_api.Connectivity
over numpy_host.NumpyHost -- ctypes + NumPy on the same C ABI), in a process of its own in which ``torch`` is never imported
(conftest.pytest_sessionfinish fails the session otherwise).  Both engines (the module's three engine selections run as usual):
goldens f1-f7, f5 / f6 / f9 / f10, f11, f13, the reference's known answers, dtypes, non-finite channels, the labelled wrapper -- what the
reference's user gets from ``pip install numpy scipy`` + this library (reference pyproject.toml:42-47)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELECTION = ("f1_cfg1 or f2_detrend or f3_every_measure or f4_lengths or f7_edges or known_answers or f5_granger or "
             "uploaded_two_sided or f6_canonical or f9_mvar or f10_global or f13_canonical or output_dtypes or "
             "nonfinite_sample or wrapper_labelled or f11_band")


def test_golden_suite_on_the_torch_free_host():
    env = dict(os.environ, SC_HIP_HOST="numpy")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-x",
                          "-p", "no:cacheprovider", "-k", SELECTION], env=env, cwd=ROOT, capture_output=True, text=True, timeout=3000)
    tail = out.stdout[-3000:] + out.stderr[-2000:]
    assert out.returncode == 0, tail
    assert " passed" in out.stdout and "torch was imported" not in out.stderr, tail
    print(out.stdout.strip().splitlines()[-1])


def test_more_than_256_signals_and_the_wrapper_on_the_torch_free_host():
    code = r"""
import sys
import numpy as np
import spectral_connectivity_amd as sc
from oracle import spectral_oracle as so
rng = np.random.default_rng(5)
T, R, C = 64, 12, 306
x = rng.standard_normal((T, R, C))
x[:, :, 1:] += 0.4 * x[:, :, :-1]
kw = dict(sampling_frequency=100.0, time_halfbandwidth_product=2)
coef, _ = so.multitaper_fft(x, fs=100.0, NW=2)
for dtype, tol in ((np.complex64, 2e-5), (np.complex128, 1e-9)):
    c = sc.Connectivity.from_multitaper(sc.Multitaper(x, **kw), dtype=dtype)
    for name in ("coherence_magnitude", "weighted_phase_lag_index", "power"):
        got, ref = getattr(c, name)(), getattr(so, name)(coef)
        ok = ~np.isnan(ref)
        assert np.array_equal(np.isnan(got), ~ok), name
        assert np.abs(got[ok] - ref[ok]).max() <= tol * max(1.0, np.abs(ref[ok]).max()), (name, dtype, np.abs(got[ok] - ref[ok]).max())
m = sc.Multitaper(x[:, :, :6], **kw)
np.testing.assert_allclose(m.fft(), so.multitaper_fft(x[:, :, :6], fs=100.0, NW=2)[0], rtol=1e-9, atol=1e-12)
res = sc.multitaper_connectivity(x[:, :, :6], sampling_frequency=100.0, time_halfbandwidth_product=2, method=["coherence_magnitude", "power"])
assert "torch" not in sys.modules, "torch was imported"
print("numpy host OK")
"""
    env = dict(os.environ, SC_HIP_HOST="numpy")
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0 and "numpy host OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_more_than_256_signals_on_the_planes_format_on_the_torch_free_host():
    """Round 6: the planes-format stage B takes more than 256 signals in one request (sc_fused2.hip plans its launches over any
    number of 32-channel blocks); on this host the whole-array planes spectra replace the channel-block tiling wherever the format
    applies (forced here by SC_PLANES_MIN_CHANNELS: the request is small), an odd count rides on its zero pad channel, and a family
    outside the format afterwards (PLV) goes back to the tiling from the series."""
    code = r"""
import sys
import numpy as np
import spectral_connectivity_amd as sc
import spectral_connectivity_amd.numpy_api as api
from oracle import spectral_oracle as so
calls = []
real = api.Connectivity._accumulate_wide
api.Connectivity._accumulate_wide = lambda self, *a, **k: (calls.append(1), real(self, *a, **k))[1]
for C in (306, 307):
    rng = np.random.default_rng(C)
    T, R = 64, 6
    x = rng.standard_normal((T, R, C))
    x[:, :, 1:] += 0.4 * x[:, :, :-1]
    x[:, :, C - 2] += 0.8 * np.roll(x[:, :, 3], 2, axis=0)
    kw = dict(sampling_frequency=100.0, time_halfbandwidth_product=2)
    coef, _ = so.multitaper_fft(x, fs=100.0, NW=2)
    c = sc.Connectivity.from_multitaper(sc.Multitaper(x, **kw), dtype=np.complex64)
    for name in ("coherence_magnitude", "weighted_phase_lag_index", "debiased_squared_weighted_phase_lag_index", "power"):
        got, ref = getattr(c, name)(), getattr(so, name)(coef)
        ok = ~np.isnan(ref)
        assert np.array_equal(np.isnan(got), ~ok), name
        assert np.abs(got[ok] - ref[ok]).max() <= 3e-5 * max(1.0, np.abs(ref[ok]).max()), (C, name, np.abs(got[ok] - ref[ok]).max())
    assert c._spectra.get("P") is not None and not calls, "the planes format was expected, without the tiling"
    got, ref = c.phase_locking_value(), so.phase_locking_value(coef)          # a family outside the format: tiled from the series
    ok = ~np.isnan(ref)
    assert np.abs(np.abs(got[ok]) - np.abs(ref[ok])).max() <= 3e-5 and calls, "PLV beyond 256 signals goes through the tiling"
    del calls[:]
assert "torch" not in sys.modules, "torch was imported"
print("numpy host OK")
"""
    env = dict(os.environ, SC_HIP_HOST="numpy", SC_PLANES_MIN_CHANNELS="44")
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0 and "numpy host OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
