"""Stage A on the anti-phase kernel (csrc/sc_mtfft_long.hip, power-of-two windows of 256 ... 4096 samples: two half-workgroups in
anti-phase, the window through half-window tiles in the exchange buffers) against the float64 oracle
(oracle/spectral_oracle.py::multitaper_fft, which follows transforms.py:1311-1405) -- every shape the kernel branches on:
channel counts around its tiles and super-tiles, odd counts, one channel, zero padding (L < N), overlapping windows, every
detrend, many trials (the engine's own choice of kernel), silent / constant / non-finite channels; and the round-3 kernels
(SC_MTFFT_LONG=0) on the same inputs.  Tolerance: the float32 engine's bar of
tests/test_gpu_parity.py, |err| <= 1e-5 |ref| + 1e-5 max |ref|."""
import numpy as np
import pytest

from oracle import spectral_oracle as so

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("kernel", ["anti-phase", "round-3"])
@pytest.mark.parametrize("N,L,step,C,R,det", [
    (2048, 2048, 2048, 1, 3, "constant"), (2048, 2048, 512, 5, 4, "linear"), (2048, 1500, 700, 16, 3, None),
    (2048, 2048, 2048, 17, 2, "constant"), (2048, 2048, 1024, 34, 3, "linear"), (2048, 2000, 2000, 130, 2, "constant"),
    (4096, 4096, 4096, 1, 2, "linear"), (4096, 4096, 1024, 3, 3, "constant"), (4096, 3000, 3000, 8, 3, None),
    (4096, 4096, 4096, 16, 2, "constant"), (4096, 4096, 2048, 18, 2, "linear"), (4096, 4000, 4000, 33, 2, "constant"),
    (4096, 4096, 4096, 66, 1, "constant"),
    (1024, 1024, 1024, 1, 3, "linear"), (1024, 700, 300, 31, 3, "constant"), (1024, 1024, 512, 32, 2, None), (1024, 1000, 1000, 70, 2, "linear"),
    (512, 512, 256, 1, 3, "constant"), (512, 300, 300, 33, 3, "linear"), (512, 512, 128, 64, 2, None), (512, 500, 250, 130, 2, "constant"),
    (256, 256, 128, 3, 4, "linear"), (256, 200, 100, 64, 3, "constant"), (256, 256, 64, 65, 3, None), (256, 256, 256, 130, 2, "linear"),
])
def test_long_windows_against_the_oracle(N, L, step, C, R, det, kernel, debug_env):
    _dev()
    debug_env("SC_MTFFT_LONG", "1" if kernel == "anti-phase" else "0")         # "1": whatever the size (few items here)
    rng = np.random.default_rng(N + 31 * C + L)
    T = L + 2 * step
    x = rng.standard_normal((T, R, C)) * (0.3 + rng.random(C)) + 4.0 * rng.standard_normal((1, R, C)) \
        + np.linspace(0, 3, T)[:, None, None] * rng.standard_normal((1, 1, C))
    got, ref = _device(x, L, step, N, det), _oracle(x, L, step, N, det)
    w = _close(got, ref, f"N={N} L={L} step={step} C={C} {det} [{kernel}]")
    print(f"\n  N={N} L={L} step={step} C={C} R={R} {det} [{kernel}]: worst err / bound {w:.2f}")


@pytest.mark.parametrize("N,C,R", [(256, 130, 200), (512, 70, 180), (1024, 70, 120), (2048, 40, 150), (4096, 24, 90)])
def test_many_trials_default_policy(N, C, R, debug_env):
    """Enough (window, trial, channel tile) items that the engine takes the anti-phase kernel by itself (no switch), overlapping
    windows, against a float64 transform of the same float32 samples; the round-3 kernels agree to float32 rounding; two runs give
    the same bits."""
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
    debug_env("SC_MTFFT_LONG", None)
    got = engine.multitaper_spectra(xd, h, L, step, N, 2, "linear").X.clone()
    again = engine.multitaper_spectra(xd, h, L, step, N, 2, "linear").X
    assert torch.equal(torch.view_as_real(got), torch.view_as_real(again))
    debug_env("SC_MTFFT_LONG", "0")
    old = engine.multitaper_spectra(xd, h, L, step, N, 2, "linear").X
    if N >= 512:        # (at 256 samples the two kernels use the same twiddles in the same order: the same bits; beyond, pass 3 differs)
        assert not torch.equal(torch.view_as_real(got), torch.view_as_real(old)), "SC_MTFFT_LONG=0 still ran the same kernel"
    xs = torch.from_numpy(x.astype(np.float64)).cuda()
    t = torch.arange(1, L + 1, dtype=torch.float64, device="cuda") / L
    A = torch.stack([t, torch.ones_like(t)], 1)
    tap = torch.from_numpy(tapers).cuda()                                             # [K, L]
    ref = []
    for w in range(2):
        seg = xs[w * step: w * step + L].reshape(L, -1)                               # [L, R * C]
        seg = (seg - A @ torch.linalg.lstsq(A, seg).solution).reshape(L, R, C)        # linear detrend, least squares
        ref.append(torch.fft.rfft(seg[None] * tap[:, :, None, None], n=N, dim=1))     # [K, F, R, C]
    ref = torch.stack(ref, 0).permute(2, 0, 3, 1, 4)                                  # [F, W, R, K, C]
    scale = ref.abs().max().item()
    for name, X in (("anti-phase", got), ("round-3", old)):
        err = (X.to(torch.complex128) - ref).abs().max().item() / scale
        print(f"\n  N={N}: {name} kernel, max |err| / max |X| against float64 = {err:.2e}")
        assert err < 2e-6, (name, err)


@pytest.mark.parametrize("N", [256, 512, 1024, 2048, 4096])
def test_silent_constant_and_nonfinite_channels_in_long_windows(N, debug_env):
    """A silent channel and a constant one (constant detrend) give EXACTLY zero coefficients, a NaN / infinity spoils its own
    channel in the windows that hold it and nothing else (transforms.py:1402-1405: every channel is transformed on its own)."""
    _dev()
    debug_env("SC_MTFFT_LONG", "1")
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


@pytest.mark.parametrize("engine_precision", ["float32", "float64"])
@pytest.mark.parametrize("N,L,step,C,R,det", [
    (8192, 8192, 4096, 5, 2, "constant"), (6000, 6000, 6000, 18, 2, "linear"), (5000, 4500, 2000, 3, 3, None),
    (448, 448, 224, 33, 3, "constant"), (2048, 2048, 1024, 6, 2, "linear"), (4096, 4000, 4000, 9, 2, "constant"),
    (308, 300, 150, 7, 3, "linear"), (130, 130, 65, 4, 3, "constant"), (50, 50, 25, 11, 4, None)])
def test_windows_through_the_taper_kernel_and_rocfft(N, L, step, C, R, det, engine_precision):
    """The lengths no fused transform takes -- beyond 4096 samples, a 7 / 11 / 13 among the factors, below 64; in the float64 engine also
    2048 / 4096 and every length outside its list -- run sc_taper_windows + rocFFT + a transposition.  Round 6: the taper kernel
    stores 16 bytes a lane (1 KB a wave and row) from 1024 samples on in float32, 512 in float64 (N % 4 == 0 / N % 2 == 0); the shorter
    windows keep the one-sample-a-lane form.  Both engines against the float64 oracle."""
    import warnings
    import spectral_connectivity_amd as sc
    from spectral_connectivity_amd import options
    _dev()
    rng = np.random.default_rng(N + 31 * C + L)
    T = L + 2 * step
    x = rng.standard_normal((T, R, C)) * (0.3 + rng.random(C)) + 4.0 * rng.standard_normal((1, R, C)) \
        + np.linspace(0, 3, T)[:, None, None] * rng.standard_normal((1, 1, C))
    ref = _oracle(x, L, step, N, det)
    old, options.precision = options.precision, engine_precision
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = sc.Multitaper(x, sampling_frequency=200.0, time_halfbandwidth_product=2.5, detrend_type=det,
                                n_time_samples_per_window=L, n_time_samples_per_step=step, n_fft_samples=N).fft()
    finally:
        options.precision = old
    if engine_precision == "float64":
        _close(got, ref, f"float64 N={N} L={L}", rtol=1e-9, atol_scale=1e-10)
    else:
        _close(got, ref, f"float32 N={N} L={L}")
