import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "default_thresholds: keep the engine's own channel thresholds for the planes format")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def pytest_generate_tests(metafunc):
    """A module that declares ``SC_PRECISIONS = ("float32", "dtype")`` runs its tests (all of them, or those named in
    ``SC_PRECISIONS_TESTS``) once per engine selection: forced
    float32 (the headline path) and the package default, where ``dtype=complex128`` -- the reference's default, what
    every test that passes no dtype gets -- selects the float64 engine."""
    precisions = getattr(metafunc.module, "SC_PRECISIONS", None)
    only = getattr(metafunc.module, "SC_PRECISIONS_TESTS", None)      # optional: the tests of the module that run under both
    if only is not None and metafunc.function.__name__ not in only:
        precisions = None
    if precisions and "_engine_precision" in metafunc.fixturenames:
        names = {"float32": "f32-engine", "float32+planes": "f32-planes"}
        metafunc.parametrize("_engine_precision", list(precisions), indirect=True,
                             ids=[names.get(p, "default-engine") for p in precisions])


@pytest.fixture(autouse=True)
def _engine_precision(request):
    """Test modules exercise the float32 engine (the headline path, north_star's tolerance) unless they declare
    ``SC_PRECISION`` (tests/test_gpu_fp64.py runs with "dtype" -- the package default) or ``SC_PRECISIONS`` (both,
    see pytest_generate_tests)."""
    from spectral_connectivity_amd import options
    old = options.precision
    want = getattr(request, "param", None) or getattr(request.module, "SC_PRECISION", "float32")
    # "float32+planes": the float32 engine with the planes format (f16 pieces, sc_fused2.hip) from two channels on -- by default
    # it starts at 32-60 channels, where the golden shapes never get
    planes = want == "float32+planes"
    old_env = os.environ.get("SC_PLANES_MIN_CHANNELS")
    if planes:
        os.environ["SC_PLANES_MIN_CHANNELS"] = "2"
    options.precision = "float32" if planes else want
    yield options.precision
    options.precision = old
    if planes:
        if old_env is None:
            os.environ.pop("SC_PLANES_MIN_CHANNELS", None)
        else:
            os.environ["SC_PLANES_MIN_CHANNELS"] = old_env


@pytest.fixture
def debug_env():
    """Set / unset one of the library's diagnostic switches (environment variables it reads once at load and again on
    sc_debug_reload_env: _lib.set_debug_env) for the rest of a test; everything is put back afterwards."""
    from spectral_connectivity_amd import _lib
    saved = {}

    def set_(name, value):
        saved.setdefault(name, os.environ.get(name))
        _lib.set_debug_env(name, value)

    yield set_
    for name, old in saved.items():
        _lib.set_debug_env(name, old)


def granger_close(got, ref, tol, what="granger"):
    """Spectral Granger predictions against a reference: entries finite in both within ``tol`` of the array maximum;
    an entry that is NaN in exactly one of the two is a prediction the reference's `gp[gp <= 0] = nan`
    (connectivity.py:1825-1848) cut on one side only -- legitimate only where the value is within ``tol`` of zero, so
    the finite one is held to that (not to a share of the entries)."""
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} != {ref.shape}"
    scale = np.nanmax(ref)
    both = ~np.isnan(got) & ~np.isnan(ref)
    err = np.abs(got[both] - ref[both]).max() if both.any() else 0.0
    assert err <= tol * scale, f"{what}: max err {err:.3e} > {tol:.1e} x {scale:.3e}"
    only_ref, only_got = ~np.isnan(ref) & np.isnan(got), np.isnan(ref) & ~np.isnan(got)
    worst = max(np.abs(ref[only_ref]).max() if only_ref.any() else 0.0, np.abs(got[only_got]).max() if only_got.any() else 0.0)
    assert worst <= tol * scale, (f"{what}: an entry that is NaN on one side only has magnitude {worst:.3e} on the other "
                                  f"(> {tol:.1e} x {scale:.3e}); {int(only_ref.sum() + only_got.sum())} one-sided NaNs")



def device_record(c, expectation_type, planes):
    """The accumulator record of ``planes`` a Connectivity object's device spectra give, as a NumPy array -- on whichever host the
    process runs (PyTorch: engine.accumulate; torch-free: the object's own record, downloaded)."""
    from spectral_connectivity_amd import _hosts
    if _hosts.kind() == "numpy":
        from spectral_connectivity_amd import numpy_api
        assert c.expectation_type == expectation_type
        _, (rec, _) = c._accumulators(planes)
        return np.array(numpy_api.host().download(rec.buf, rec.shape, np.float64 if rec.f64 else np.float32))
    import torch
    from spectral_connectivity_amd import engine
    accum, _ = engine.accumulate(c._device(), expectation_type, planes, n_freq=c._n_freq)
    torch.cuda.synchronize()
    return accum.cpu().numpy()


def pytest_sessionfinish(session, exitstatus):
    """SC_HIP_HOST=numpy: the whole session must have run without torch (the reference depends on NumPy + SciPy only)."""
    if os.environ.get("SC_HIP_HOST") == "numpy" and "torch" in sys.modules:
        print("\nERROR: SC_HIP_HOST=numpy but torch was imported during the session", file=sys.stderr)
        session.exitstatus = 1


def unpack_record_planes(accum, C):
    """Accumulator records [n_bins, n_planes * n_tiles * 256] (include/sc_hip.h: upper-triangular 16 x 16 channel tiles,
    tile index = bi NB - bi (bi - 1) / 2 + (bj - bi)) -> [n_bins, n_planes, C, C] float64, NaN where no tile holds the
    entry (below the diagonal tiles)."""
    rec = np.asarray(accum, dtype=np.float64)
    n_bins = rec.shape[0]
    NB = (C + 15) // 16
    n_tiles = NB * (NB + 1) // 2
    n_planes = rec.shape[1] // (n_tiles * 256)
    tiles = rec.reshape(n_bins, n_planes, n_tiles, 16, 16)
    out = np.full((n_bins, n_planes, NB * 16, NB * 16), np.nan)
    for bi in range(NB):
        for bj in range(bi, NB):
            t = bi * NB - bi * (bi - 1) // 2 + (bj - bi)
            out[:, :, bi * 16:(bi + 1) * 16, bj * 16:(bj + 1) * 16] = tiles[:, :, t]
    return out[:, :, :C, :C]
