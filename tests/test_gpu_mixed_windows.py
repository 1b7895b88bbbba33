"""Stage A for the window lengths that are not powers of two on the register-resident kernel (csrc/sc_mtfft_mixed.hip: N = 10 RM RF =
100 ... 2000 samples, radix-10 first pass in registers, two exchanges, anti-phase half-workgroups, planes output) against the float64
oracle (oracle/spectral_oracle.py::multitaper_fft, which follows transforms.py:1311-1405; n_fft = next_fast_len(L), transforms.py:
1024-1036) -- every shape the kernel branches on: one channel, odd counts, counts around its channel tiles and super-tiles, zero
padding (L < N), overlapping windows, every detrend, silent / constant / non-finite channels, every geometry it is built with --
and the round-2 kernels (SC_MTFFT_MIXED=0) on the same inputs.  Tolerance: the float32 engine's bar of tests/test_gpu_parity.py,
|err| <= 1e-5 |ref| + 1e-5 max |ref|; against a float64 transform of the same float32 samples 2e-6 of the largest coefficient."""
import numpy as np
import pytest

from oracle import spectral_oracle as so

pytestmark = pytest.mark.gpu
LENGTHS = (100, 150, 160, 200, 240, 250, 300, 320, 360, 400, 450, 480, 500, 600, 750, 800, 900, 1000, 1200, 1250, 1500, 1600, 1800, 2000)


def _dev():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    from spectral_connectivity_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _close(got, ref, what, rtol=1e-5, atol_scale=1e-5):
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    nan_g, nan_r = np.isnan(got), np.isnan(ref)
    assert np.array_equal(nan_g, nan_r), f"{what}: NaN pattern differs ({nan_g.sum()} vs {nan_r.sum()})"
    ok = ~nan_r
    scale = np.abs(ref[ok]).max()
    worst = (np.abs(got[ok] - ref[ok]) / (rtol * np.abs(ref[ok]) + atol_scale * scale)).max()
    assert worst <= 1.0, f"{what}: worst err / bound {worst:.2f}"
    return worst


def _oracle(x, L, step, N, det, NW=2.5, fs=200.0):
    return so.multitaper_fft(x, fs=fs, NW=NW, detrend_type=det, n_time_samples_per_window=L, n_time_samples_per_step=step,
                             n_fft_samples=N)[0]


def _device(x, L, step, N, det, NW=2.5, fs=200.0):
    """Two-sided coefficients [W, R, K, N, C] through the public class, float32 engine."""
    import warnings
    import spectral_connectivity_amd as sc
    from spectral_connectivity_amd import options
    old, options.precision = options.precision, "float32"
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = sc.Multitaper(x, sampling_frequency=fs, time_halfbandwidth_product=NW, detrend_type=det,
                              n_time_samples_per_window=L, n_time_samples_per_step=step, n_fft_samples=N)
            return m.fft()
    finally:
        options.precision = old


@pytest.mark.parametrize("kernel", ["register passes", "round-2"])
@pytest.mark.parametrize("N,L,step,C,R,det", [
    (200, 200, 100, 1, 3, "constant"), (200, 200, 200, 47, 2, "linear"), (200, 150, 70, 48, 3, None), (200, 200, 50, 49, 2, "constant"),
    (250, 250, 125, 3, 4, "linear"), (250, 250, 250, 32, 2, "constant"), (250, 180, 90, 33, 3, None), (250, 250, 125, 130, 2, "linear"),
    (300, 300, 300, 31, 2, "constant"), (300, 256, 128, 66, 2, "linear"),
    (400, 400, 200, 17, 2, None), (400, 333, 333, 34, 2, "constant"),
    (500, 500, 250, 1, 3, "linear"), (500, 500, 500, 16, 2, "constant"), (500, 400, 100, 35, 2, None), (500, 500, 250, 70, 2, "linear"),
    (600, 600, 300, 18, 2, "constant"), (600, 512, 512, 33, 2, "linear"),
    (750, 750, 375, 15, 2, "constant"), (750, 700, 700, 34, 2, None),
    (800, 800, 400, 16, 2, "linear"), (800, 640, 320, 21, 2, "constant"),
    (1000, 1000, 1000, 1, 2, "constant"), (1000, 1000, 500, 16, 2, "linear"), (1000, 900, 450, 17, 2, None),
    (1000, 1000, 1000, 33, 2, "constant"), (1000, 1000, 250, 66, 1, "linear"),
    (1200, 1200, 600, 14, 2, "constant"), (1200, 1100, 1100, 20, 2, "linear"),
    (1500, 1500, 750, 9, 2, None), (1500, 1400, 700, 18, 2, "constant"),
    (2000, 2000, 1000, 7, 2, "linear"), (2000, 1900, 1900, 10, 2, "constant"), (2000, 2000, 2000, 34, 1, None),
    (100, 100, 50, 97, 2, "constant"), (100, 80, 80, 5, 3, "linear"), (150, 150, 75, 33, 2, None), (160, 160, 160, 64, 2, "linear"),
    (240, 240, 120, 17, 2, "constant"), (320, 300, 150, 34, 2, "linear"), (360, 360, 360, 9, 2, None), (450, 450, 225, 18, 2, "constant"),
    (480, 480, 240, 25, 2, "linear"), (900, 900, 450, 10, 2, "constant"), (1250, 1250, 625, 6, 2, "linear"),
    (1600, 1500, 1500, 7, 2, None), (1800, 1800, 900, 5, 2, "constant"),
])
def test_mixed_windows_against_the_oracle(N, L, step, C, R, det, kernel, debug_env):
    _dev()
    debug_env("SC_MTFFT_MIXED", "1" if kernel == "register passes" else "0")      # "1": whatever the size (few items here)
    rng = np.random.default_rng(N + 31 * C + L)
    T = L + 2 * step
    x = rng.standard_normal((T, R, C)) * (0.3 + rng.random(C)) + 4.0 * rng.standard_normal((1, R, C)) \
        + np.linspace(0, 3, T)[:, None, None] * rng.standard_normal((1, 1, C))
    got, ref = _device(x, L, step, N, det), _oracle(x, L, step, N, det)
    w = _close(got, ref, f"N={N} L={L} step={step} C={C} {det} [{kernel}]")
    print(f"\n  N={N} L={L} step={step} C={C} R={R} {det} [{kernel}]: worst err / bound {w:.2f}")


def _float64_transform(x, tapers, L, step, N, W, R, C):
    """rfft of the linearly detrended, tapered windows in float64 torch on the device: [F, W, R, K, C]."""
    import torch
    xs = torch.from_numpy(x.astype(np.float64)).cuda()
    t = torch.arange(1, L + 1, dtype=torch.float64, device="cuda") / L
    A = torch.stack([t, torch.ones_like(t)], 1)
    tap = torch.from_numpy(tapers).cuda()                                             # [K, L]
    ref = []
    for w in range(W):
        seg = xs[w * step: w * step + L].reshape(L, -1)                               # [L, R * C]
        seg = (seg - A @ torch.linalg.lstsq(A, seg).solution).reshape(L, R, C)        # linear detrend, least squares
        ref.append(torch.fft.rfft(seg[None] * tap[:, :, None, None], n=N, dim=1))     # [K, F, R, C]
    return torch.stack(ref, 0).permute(2, 0, 3, 1, 4)


@pytest.mark.parametrize("N,C,R", [(200, 130, 200), (250, 70, 300), (500, 70, 200), (1000, 40, 200), (2000, 24, 160)])
def test_many_trials_default_policy(N, C, R, debug_env):
    """Enough (window, trial, channel tile) items that the engine takes the register-resident kernel by itself (no switch),
    overlapping windows, against a float64 transform of the same float32 samples; the round-2 kernels agree to float32 rounding (and
    are a different kernel: different bits); two runs give the same bits."""
    import torch
    from spectral_connectivity_amd import engine
    from spectral_connectivity_amd.transforms import dpss_windows
    _dev()
    rng = np.random.default_rng(N + C)
    L, step = N, N // 2
    T = L + step
    x = (rng.standard_normal((T, R, C)) + 2.0).astype(np.float32)
    tapers = np.asarray(dpss_windows(L, 2.0, 3)[0], dtype=np.float64)
    xd, h = torch.from_numpy(x).cuda(), torch.from_numpy(np.ascontiguousarray(tapers, dtype=np.float32)).cuda()
    debug_env("SC_MTFFT_MIXED", None)
    got = engine.multitaper_spectra(xd, h, L, step, N, 2, "linear").X.clone()
    again = engine.multitaper_spectra(xd, h, L, step, N, 2, "linear").X
    assert torch.equal(torch.view_as_real(got), torch.view_as_real(again))
    debug_env("SC_MTFFT_MIXED", "0")
    old = engine.multitaper_spectra(xd, h, L, step, N, 2, "linear").X
    assert not torch.equal(torch.view_as_real(got), torch.view_as_real(old)), "SC_MTFFT_MIXED=0 still ran the same kernel"
    ref = _float64_transform(x, tapers, L, step, N, 2, R, C)
    scale = ref.abs().max().item()
    for name, X in (("register passes", got), ("round-2", old)):
        err = (X.to(torch.complex128) - ref).abs().max().item() / scale
        print(f"\n  N={N}: {name} kernel, max |err| / max |X| against float64 = {err:.2e}")
        assert err < 2e-6, (name, err)


@pytest.mark.parametrize("N", LENGTHS)
def test_every_geometry_of_a_length(N, debug_env):
    """The workgroup geometries a length is built with (threads of a half, lanes aligned to waves or packed: SC_MTFFT_MIXED_GEO) are
    the same arithmetic in another arrangement: both outputs, against the float64 transform, and the planes output decoded (every
    channel to 1.5e-6 of ITS OWN largest coefficient beside a partner 200 times louder)."""
    import os
    import torch
    from spectral_connectivity_amd import _lib, engine
    from spectral_connectivity_amd.transforms import dpss_windows
    _dev()
    os.environ["SC_PLANES_MIN_CHANNELS"] = "2"
    try:
        rng = np.random.default_rng(N)
        C, R, L, step = 38, 3, N - 6, N // 2
        T = L + 2 * step
        W = (T - L) // step + 1
        x = (rng.standard_normal((T, R, C)) + 1.5).astype(np.float32)
        x[:, :, 4] *= 200.0
        x[:, :, 5] *= 5e-3
        tapers = np.asarray(dpss_windows(L, 2.0, 3)[0], dtype=np.float64)
        xd, h = torch.from_numpy(x).cuda(), torch.from_numpy(np.ascontiguousarray(tapers, dtype=np.float32)).cuda()
        ref = _float64_transform(x, tapers, L, step, N, W, R, C)
        amax = ref.abs().amax(dim=(0, 1, 2, 3))
        debug_env("SC_MTFFT_MIXED", "1")
        seen = []
        for geo in ("0", "1", "2", "3"):
            debug_env("SC_MTFFT_MIXED_GEO", geo)
            X = engine.multitaper_spectra(xd, h, L, step, N, W, "linear").X
            sp = engine.multitaper_spectra(xd, h, L, step, N, W, "linear", planes_hint=_lib.PLANE_CSM | _lib.PLANE_ABS_IM)
            assert sp.P is not None and sp._X is None, "the planes format was expected for this length"
            e1 = ((X.to(torch.complex128) - ref).abs().amax(dim=(0, 1, 2, 3)) / amax).max().item()
            e2 = ((sp.X.to(torch.complex128) - ref).abs().amax(dim=(0, 1, 2, 3)) / amax).max().item()
            seen.append((geo, e1, e2))
            assert e1 < 1.5e-6 and e2 < 1.5e-6, seen
        print(f"\n  N={N}: (geometry, complex64 err, planes err) {seen}")
    finally:
        os.environ.pop("SC_PLANES_MIN_CHANNELS", None)


@pytest.mark.parametrize("N", [200, 250, 500, 1000, 1500, 2000])
def test_silent_constant_and_nonfinite_channels_in_mixed_windows(N, debug_env):
    """A silent channel and a constant one (constant detrend) give EXACTLY zero coefficients, a NaN / infinity spoils its own
    channel in the windows that hold it and nothing else (transforms.py:1402-1405: every channel is transformed on its own)."""
    _dev()
    debug_env("SC_MTFFT_MIXED", "1")
    rng = np.random.default_rng(N)
    C, R, L, step = 10, 3, N, N // 2
    T = L + 2 * step
    x = rng.standard_normal((T, R, C)) * 2.0 + 1.0
    x[:, :, 2] = 0.0
    x[:, :, 7] = 0.75
    x[L // 3, 1, 4] = np.nan              # windows 0 of trial 1, channel 4 (partner: channel 5)
    x[L + step + 5, 2, 9] = np.inf        # window 2 only of trial 2, channel 9 (partner: channel 8)
    clean = np.where(np.isfinite(x), x, 0.0)
    got, ref = _device(x, L, step, N, "constant"), _oracle(clean, L, step, N, "constant")
    bad = np.zeros(got.shape, dtype=bool)
    bad[0, 1, :, :, 4] = True
    bad[2, 2, :, :, 9] = True
    assert np.isnan(got[bad]).all() and np.isfinite(got[~bad]).all()
    assert np.all(got[..., 2] == 0) and np.all(got[..., 7] == 0)
    _close(np.where(bad, 0, got), np.where(bad, 0, ref), f"N={N}: channels beside a silent / non-finite one")


@pytest.mark.parametrize("L,C", [(250, 64), (1000, 48), (200, 130), (500, 70)])
def test_public_classes_on_the_planes_format_at_these_lengths(L, C, monkeypatch):
    """Multitaper -> Connectivity.from_multitaper(dtype=complex64) at window lengths that are not powers of two: stage A writes the
    planes format (sc_mtfft_mixed.hip: the only kernel with that output there), stage B is sc_fused2.hip on the f16 pieces --
    asserted: the spectra are never decoded --, coherence / imaginary coherence / wPLI / power against the float64 oracle at the
    float32 engine's bar, and the same numbers as with SC_PLANES_FORMAT=0 (complex64 spectra, the round-3 kernels) to that bar."""
    import spectral_connectivity_amd as sc
    _dev()
    monkeypatch.setenv("SC_PLANES_MIN_CHANNELS", "2")
    rng = np.random.default_rng(L + C)
    R, step = 6, L // 2
    T = L + 3 * step
    t = np.arange(T) / 1000.0
    x = rng.standard_normal((T, R, C))
    x += 0.6 * np.sin(2 * np.pi * 60 * t[:, None, None] + 2 * np.pi * np.arange(C)[None, None, :] / C)
    kw = dict(sampling_frequency=1000.0, time_halfbandwidth_product=3, n_time_samples_per_window=L, n_time_samples_per_step=step)
    coef, _ = so.multitaper_fft(x, fs=1000.0, NW=3, n_time_samples_per_window=L, n_time_samples_per_step=step)
    c = sc.Connectivity.from_multitaper(sc.Multitaper(x.astype(np.float32), **kw), dtype=np.complex64)
    got = {name: getattr(c, name)() for name in ("coherence_magnitude", "weighted_phase_lag_index", "imaginary_coherence", "power")}
    assert c._spectra.P is not None and c._spectra._X is None, "the planes format (never decoded) was expected"
    for name, g in got.items():
        w = _close(g, getattr(so, name)(coef), f"L={L} C={C} {name}")
        print(f"\n  L={L} C={C} {name}: worst err / bound {w:.2f}")
