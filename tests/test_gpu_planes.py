"""Planes format (two f16 pieces per real number, sc_fused2.hip / sc_multitaper_fft_planes_f32): the device format the float32
engine uses for CSM (+ |Im s|) accumulators of up to 128 signals.  Stage A into the format against the complex64 transform, the
format's conversions, and stage B on it against the CPU oracle (reference connectivity.py:447-526, :982-1028) -- through the C
ABI and through the public classes."""
from ctypes import byref

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from oracle import spectral_oracle as so                                     # noqa: E402
from spectral_connectivity_amd import Connectivity, Multitaper, _lib, engine, transforms   # noqa: E402

PLANES = _lib.PLANE_CSM | _lib.PLANE_ABS_IM


@pytest.fixture(autouse=True)
def _planes_from_two_channels(monkeypatch, request):
    """The engine takes the planes format from 32-60 channels on (below, the small-channel kernel on complex64 spectra is
    faster); the tests of the format itself lift that threshold so that small shapes exercise it too."""
    if "default_thresholds" not in request.keywords:
        monkeypatch.setenv("SC_PLANES_MIN_CHANNELS", "2")


def _dev():
    _lib.require_gpu()
    return torch.device("cuda:0")


def _series(T, R, C, seed, offset=0.0, quiet=2e-3, loud=250.0):
    rng = np.random.default_rng(seed)
    t = np.arange(T) / 1000.0
    x = rng.standard_normal((T, R, C))
    # a shared 60 Hz component with a different lag in every channel (a zero-lag copy would make Im s a difference of large
    # numbers: the float32 engine's rounding, whatever the device format, would then dominate the phase-lag measures)
    x += 0.6 * np.sin(2 * np.pi * 60 * t[:, None, None] + 2 * np.pi * np.arange(C)[None, None, :] / C)
    x[:, :, 0] *= loud                       # a loud channel
    x[:, :, C - 1] *= quiet                  # a quiet one
    return x + offset


@pytest.mark.parametrize("kernel", ["engine's choice", "anti-phase"])
@pytest.mark.parametrize("L,step,C,detrend", [(256, 128, 128, "constant"), (128, 64, 20, "linear"), (64, 64, 34, None),
                                               (512, 256, 16, "constant"), (1024, 1024, 48, "constant"), (256, 256, 70, "linear"),
                                               (512, 512, 100, None), (1024, 512, 34, "linear"), (2048, 2048, 48, "constant"),
                                               (2048, 1024, 22, "linear"), (4096, 4096, 20, "constant"), (4096, 2048, 36, None),
                                               # the lengths that are not powers of two (sc_mtfft_mixed.hip: the only kernel with this
                                               # output there, both values of `kernel` run it)
                                               (250, 125, 128, "constant"), (200, 100, 20, "linear"), (500, 250, 34, None),
                                               (1000, 1000, 48, "constant"), (1000, 500, 22, "linear"), (300, 300, 70, "constant"),
                                               (400, 200, 36, None), (600, 300, 24, "linear"), (750, 750, 18, "constant"),
                                               (800, 400, 40, None), (1200, 1200, 16, "linear"), (1500, 750, 12, "constant"),
                                               (2000, 2000, 10, "linear"),
                                               # every other 2^a 3^b 5^c length (round 6: the planes store of the Stockham kernel in
                                               # sc_mtfft.hip; last tiles of 2 ... 22 channels zeroed by the kernel itself)
                                               (384, 192, 34, "constant"), (768, 768, 20, "linear"), (96, 48, 70, None),
                                               (1536, 1536, 12, "constant"), (960, 480, 22, "linear"), (108, 54, 36, "constant")])
def test_stage_a_planes_decode_to_the_spectra(L, step, C, detrend, kernel, debug_env):
    """sc_multitaper_fft_planes_f32 + sc_spectra_from_planes_f32 against the float64 transform of the same samples and
    tapers: the float32 transform's rounding plus the 22 bits of the format.  The scales sit on the SAMPLES, so the two
    channels that share a complex transform enter it at the same magnitude: a channel far weaker than its pair partner (here
    x 250 and x 1/500) comes out to its OWN rounding (the complex64 kernel normalises the pair per window for the same effect;
    before round 4 it left the partner's rounding on the weak channel: 6e-5 of its largest coefficient)."""
    dev, lib = _dev(), _lib.load()
    # ("anti-phase": sc_mtfft_long.hip from 256 samples on, whatever the size; the engine's choice at these sizes: the round-1..3
    #  kernels up to 1024 samples, the anti-phase kernel -- the only one with this output there -- beyond)
    debug_env("SC_MTFFT_LONG", "1" if kernel == "anti-phase" else None)
    T, R, NW = max(2048, 2 * L), 3, 3
    K = 2 * NW - 1
    W = (T - L) // step + 1
    x = torch.from_numpy(_series(T, R, C, seed=L + C).astype(np.float32)).to(dev)
    tapers = np.asarray(transforms.dpss_windows(L, NW, K)[0])[:K]
    h = torch.from_numpy(np.ascontiguousarray(tapers * np.sqrt(1000.0) / 1000.0, dtype=np.float32)).to(dev)
    X64 = engine.multitaper_spectra_f64(x.double(), h.double(), L, step, L, W, detrend).X
    ref = engine.multitaper_spectra(x, h, L, step, L, W, detrend)
    sp = engine.multitaper_spectra(x, h, L, step, L, W, detrend, planes_hint=PLANES)
    assert sp.P is not None and sp._X is None, "the planes format was expected for this shape"
    X, Xr = sp.X, ref.X
    scaled_max = (X64.abs() * sp.scale[:C]).max().item()
    assert scaled_max < 32768.0, "a coefficient left the range the scales promise"
    amax = X64.abs().amax(dim=(0, 1, 2, 3))
    err = ((X - X64).abs().amax(dim=(0, 1, 2, 3)) / amax)
    err_c64 = ((Xr - X64).abs().amax(dim=(0, 1, 2, 3)) / amax)
    # every channel to (a few ulp of float32) x its own largest coefficient -- the pair partner's size does not enter, in either
    # device format (complex64: the two channels of a pair are scaled to [1, 2) per window inside the kernel and scaled back)
    assert err.max().item() < 1.5e-6, (err.max().item(), err_c64.max().item())
    assert err_c64.max().item() < 1.5e-6, (err.max().item(), err_c64.max().item())


@pytest.mark.parametrize("L,step,C", [(250, 125, 12), (1000, 1000, 6), (200, 100, 40), (2048, 2048, 6)])
def test_complex64_transform_keeps_a_weak_channel_from_its_pair_partners_rounding(L, step, C):
    """The complex64 kernels (radix-16, long windows and the mixed-radix lengths next_fast_len hands out) normalise the two
    channels of a packed pair per window: every channel within a few float32 ulp of ITS OWN largest coefficient, next to a
    partner 250 times louder or 500 times quieter."""
    dev = _dev()
    T, R, NW = 4096, 2, 2
    K = 2 * NW - 1
    W = (T - L) // step + 1
    x = torch.from_numpy(_series(T, R, C, seed=L + C).astype(np.float32)).to(dev)
    tapers = np.asarray(transforms.dpss_windows(L, NW, K)[0])[:K]
    h = torch.from_numpy(np.ascontiguousarray(tapers * np.sqrt(1000.0) / 1000.0, dtype=np.float32)).to(dev)
    X64 = engine.multitaper_spectra_f64(x.double(), h.double(), L, step, L, W, "constant").X
    X = engine.multitaper_spectra(x, h, L, step, L, W, "constant").X
    amax = X64.abs().amax(dim=(0, 1, 2, 3))
    err = (X - X64).abs().amax(dim=(0, 1, 2, 3)) / amax
    assert err.max().item() < 2.5e-6, err.tolist()


def test_conversions_are_inverse_and_keep_zero_and_nonfinite_channels():
    dev, lib = _dev(), _lib.load()
    F, W, R, K, C = 3, 2, 4, 3, 36
    g = torch.Generator(device=dev).manual_seed(5)
    X = torch.view_as_complex(torch.randn((F, W, R, K, C, 2), dtype=torch.float32, device=dev, generator=g))
    X[..., 7] = 0
    X[..., 9] *= 1e-12
    X[..., 11] *= 1e9
    sp = engine.DeviceSpectra(X, (F, W, R, K, C), (W * R * K * C, R * K * C, K * C, C), 4, True, C_alloc=C)
    d = sp.desc("trials_tapers")
    P = torch.zeros((F * W * R * K * lib.sc_planes_row_bytes(C),), dtype=torch.uint8, device=dev)
    scale = torch.empty((2 * C,), dtype=torch.float32, device=dev)
    work = torch.empty((C,), dtype=torch.int32, device=dev)
    _lib.check(lib.sc_planes_scales_from_spectra_f32(X.data_ptr(), F * W * R * K, C, scale.data_ptr(), work.data_ptr(), None), "scales")
    _lib.check(lib.sc_planes_from_spectra_f32(X.data_ptr(), byref(d), scale.data_ptr(), P.data_ptr(), None), "to planes")
    Xb = torch.full_like(X, 7.0)
    _lib.check(lib.sc_spectra_from_planes_f32(P.data_ptr(), byref(d), scale.data_ptr(), Xb.data_ptr(), None), "from planes")
    torch.cuda.synchronize()
    s = scale[:C]
    assert bool((torch.log2(s) == torch.log2(s).round()).all()) and bool((s * scale[C:] == 1).all()), "scales are powers of two"
    assert float(s[7]) == 1.0 and bool((Xb[..., 7] == 0).all())
    rel = ((Xb - X).abs() / X.abs().clamp_min(1e-38))
    assert rel[X.abs() > 0].max().item() < 2.5e-7


@pytest.mark.parametrize("C,R,K,expectation", [(128, 12, 7, "trials_tapers"), (100, 90, 7, "trials_tapers"), (20, 3, 3, "trials_tapers"),
                                                (64, 40, 5, "trials"), (34, 6, 7, "tapers"), (32, 30, 3, "time_trials"), (6, 1, 1, "time")])
def test_stage_b_on_planes_against_the_oracle(C, R, K, expectation):
    """The records of sc_fused2_csm_absim_f32 -- read back as E[s] (cross-spectral matrix) and E[Im s] / E[|Im s|] (wPLI) --
    against float64 NumPy sums of the per-observation cross-spectra (reference connectivity.py:463-526, :982-1028), for
    channel scales spread over six decades and every expectation type whose observations form one run of rows."""
    from spectral_connectivity_amd.connectivity import EXPECTATION_AXES
    dev, lib = _dev(), _lib.load()
    W, N = 3, 16
    F = N // 2 + 1
    rng = np.random.default_rng(C * 7 + R)
    coef = rng.standard_normal((W, R, K, F, C)) + 1j * rng.standard_normal((W, R, K, F, C))
    coef = coef + 0.4 * coef[..., :1]                                  # a shared component (before the scales: no cancelling pairs)
    coef = coef * (0.2 + rng.random(C)) * 10.0 ** rng.integers(-3, 4, C)
    X = torch.from_numpy(np.ascontiguousarray(np.moveaxis(coef, 3, 0)).astype(np.complex64)).to(dev)     # [F][W][R][K][C]
    strides = (W * R * K * C, R * K * C, K * C, C)
    sp = engine.DeviceSpectra(X, (F, W, R, K, C), strides, N, True, C_alloc=C)
    assert lib.sc_fused2_supported(byref(sp.desc(expectation)), PLANES)
    P = torch.zeros((F * W * R * K * lib.sc_planes_row_bytes(C),), dtype=torch.uint8, device=dev)
    scale = torch.empty((2 * C,), dtype=torch.float32, device=dev)
    work = torch.empty((C,), dtype=torch.int32, device=dev)
    _lib.check(lib.sc_planes_scales_from_spectra_f32(X.data_ptr(), F * W * R * K, C, scale.data_ptr(), work.data_ptr(), None), "scales")
    _lib.check(lib.sc_planes_from_spectra_f32(X.data_ptr(), byref(sp.desc("trials_tapers")), scale.data_ptr(), P.data_ptr(), None), "to planes")
    spp = engine.DeviceSpectra(None, (F, W, R, K, C), strides, N, True, C_alloc=C, P=P, scale=scale)
    accum, n_obs = engine.accumulate(spp, expectation, PLANES)
    csm = engine.to_host(engine.measure(accum, C, PLANES, n_obs, _lib.M_CSM)).reshape(-1, F, C, C)
    wpli = engine.to_host(engine.measure(accum, C, PLANES, n_obs, _lib.M_WPLI)).reshape(-1, F, C, C)
    c = np.moveaxis(X.cpu().numpy().astype(np.complex128), 0, 3)              # (W, R, K, F, C): the complex64-rounded input
    axes = tuple(EXPECTATION_AXES[expectation])
    s = c[..., :, None] * np.conj(c[..., None, :])
    S = s.mean(axis=axes).reshape(-1, F, C, C)
    A = np.abs(s.imag).mean(axis=axes).reshape(-1, F, C, C)
    assert n_obs == int(np.prod([(W, R, K)[a] for a in axes]))
    P_ = np.sqrt(np.real(np.einsum("gfii->gfi", S)))
    scale_ij = P_[..., :, None] * P_[..., None, :]
    assert np.max(np.abs(csm - S) / scale_ij) < 4e-6              # relative to sqrt(P_i P_j): the f32 accumulation's own level
    ref_wpli = S.imag / np.where(A < np.finfo(float).eps, 1.0, A)
    off = ~np.eye(C, dtype=bool)
    assert np.allclose(wpli[..., off], ref_wpli[..., off], rtol=1e-5, atol=4e-6)


@pytest.mark.parametrize("dtype", [np.complex64])
def test_public_classes_take_the_planes_path_and_match_the_oracle(dtype):
    """Multitaper -> Connectivity(dtype=complex64) on a shape the planes format applies to: coherence and wPLI equal the CPU
    oracle's, the spectra were held as f16 pieces, and a measure outside the format's reach (PLV) decodes them transparently."""
    _dev()
    # (channel amplitudes within 1 : 8 : 0.1 here: stage A packs two real channels into one complex transform, so a channel hundreds of times weaker than its
    #  pair partner would carry the partner's float32 rounding -- a property of the float32 engine in either device format)
    x = _series(1024, 9, 12, seed=3, quiet=0.1, loud=8.0)
    m = Multitaper(x, sampling_frequency=1000, time_halfbandwidth_product=3, n_time_samples_per_window=256, n_time_samples_per_step=128)
    c = Connectivity.from_multitaper(m, dtype=dtype)
    coh, wpli = c.coherence_magnitude(), c.weighted_phase_lag_index()
    assert c._spectra.P is not None and c._spectra._X is None
    coef, _ = so.multitaper_fft(x, fs=1000, NW=3, n_time_samples_per_window=256, n_time_samples_per_step=128)
    np.testing.assert_allclose(coh, so.coherence_magnitude(coef), rtol=2e-4, atol=2e-5, equal_nan=True)
    np.testing.assert_allclose(wpli, so.weighted_phase_lag_index(coef), rtol=2e-4, atol=2e-5)
    # the complex64 path gives the same numbers
    import os
    os.environ["SC_PLANES_FORMAT"] = "0"
    try:
        c64 = Connectivity.from_multitaper(Multitaper(x, sampling_frequency=1000, time_halfbandwidth_product=3,
                                                      n_time_samples_per_window=256, n_time_samples_per_step=128), dtype=dtype)
        coh64, wpli64 = c64.coherence_magnitude(), c64.weighted_phase_lag_index()
        assert c64._spectra.P is None
    finally:
        del os.environ["SC_PLANES_FORMAT"]
    np.testing.assert_allclose(coh, coh64, rtol=0, atol=2e-6, equal_nan=True)
    np.testing.assert_allclose(wpli, wpli64, rtol=0, atol=2e-6)
    plv = c.phase_locking_value()
    assert c._spectra._X is not None
    np.testing.assert_allclose(plv, so.phase_locking_value(coef), rtol=2e-4, atol=2e-5, equal_nan=True)


@pytest.mark.default_thresholds
def test_when_the_engine_takes_the_planes_format():
    """Round 5: the device format is a function of the SHAPE of the request for every accumulator family the planes kernels
    serve -- 44 ... 256 signals, at least 256 MB of spectra, a window the planes transform takes -- not of which family is asked
    for first; the unit-phasor family and hint-less callers keep complex64."""
    big = 1 << 30
    csm = _lib.PLANE_CSM
    for fam in _lib.PLANES_FORMAT_FAMILIES:
        assert _lib.planes_format_applies(256, 256, 64, fam, spectra_bytes=big)
        assert _lib.planes_format_applies(256, 256, 128, fam, spectra_bytes=big)
        assert _lib.planes_format_applies(1024, 1024, 256, fam, spectra_bytes=big)
        assert not _lib.planes_format_applies(256, 256, 40, fam, spectra_bytes=big)
        assert not _lib.planes_format_applies(256, 256, 64, fam, spectra_bytes=32 << 20)
        assert _lib.planes_format_applies(2048, 2048, 64, fam, spectra_bytes=big)          # (round 5: 2048 and 4096 samples too)
        assert _lib.planes_format_applies(4096, 4096, 64, fam, spectra_bytes=big)
        assert not _lib.planes_format_applies(8192, 8192, 64, fam, spectra_bytes=big)
        assert _lib.planes_format_applies(256, 256, 258, fam, spectra_bytes=big)           # (round 6: up to 1024 signals)
        assert not _lib.planes_format_applies(256, 256, 1026, fam, spectra_bytes=big)
        assert _lib.planes_format_applies(384, 384, 64, fam, spectra_bytes=big)             # (round 6: every 2^a 3^b 5^c <= 2048)
        assert not _lib.planes_format_applies(448, 448, 64, fam, spectra_bytes=big)         # (7 among the factors)
    assert not _lib.planes_format_applies(256, 256, 64, _lib.PLANE_CSM | _lib.PLANE_UNIT, spectra_bytes=big)
    assert not _lib.planes_format_applies(256, 256, 64, _lib.PLANE_UNIT, spectra_bytes=big)
    assert not _lib.planes_format_applies(256, 256, 64, None, spectra_bytes=big)
    dev = _dev()
    tapers = np.asarray(transforms.dpss_windows(128, 2, 3)[0])[:3]
    h = torch.from_numpy(np.ascontiguousarray(tapers / np.sqrt(1000.0), dtype=np.float32)).to(dev)
    x = torch.from_numpy(_series(512, 2, 64, seed=1).astype(np.float32)).to(dev)
    sp = engine.multitaper_spectra(x, h, 128, 128, 128, 4, "constant", planes_hint=PLANES)        # 0.4 MB of spectra
    assert sp.P is None
    sp = engine.multitaper_spectra(x, h, 128, 128, 128, 4, "constant", planes_hint=csm)
    assert sp.P is None


def test_planes_switch_gives_the_complex64_path():
    """SC_PLANES_FORMAT=0: the same call keeps complex64 spectra (the format is a device detail, not an interface)."""
    import os
    dev = _dev()
    x = torch.from_numpy(_series(512, 2, 8, seed=1).astype(np.float32)).to(dev)
    tapers = np.asarray(transforms.dpss_windows(128, 2, 3)[0])[:3]
    h = torch.from_numpy(np.ascontiguousarray(tapers / np.sqrt(1000.0), dtype=np.float32)).to(dev)
    os.environ["SC_PLANES_FORMAT"] = "0"
    try:
        sp = engine.multitaper_spectra(x, h, 128, 128, 128, 4, "constant", planes_hint=PLANES)
    finally:
        del os.environ["SC_PLANES_FORMAT"]
    assert sp.P is None and sp._X is not None


@pytest.mark.parametrize("C", [64, 130, 200])
def test_every_planes_family_through_the_public_classes(C):
    """coherence, wPLI, debiased wPLI and PLI of 64 ... 200 channels (above 128: several launches over 32-channel blocks;
    (Im s)^2 and sign(Im s): plane passes of the same kernel) through Multitaper -> Connectivity(dtype=complex64), every one
    computed from spectra held as f16 pieces, against the CPU oracle."""
    _dev()
    R, L = 5, 64
    x = _series(192, R, C, seed=C, quiet=0.2, loud=4.0)
    kw = dict(sampling_frequency=1000, time_halfbandwidth_product=2, n_time_samples_per_window=L, n_time_samples_per_step=L)
    coef, _ = so.multitaper_fft(x, fs=1000, NW=2, n_time_samples_per_window=L, n_time_samples_per_step=L)
    checks = (("coherence_magnitude", so.coherence_magnitude), ("weighted_phase_lag_index", so.weighted_phase_lag_index),
              ("debiased_squared_weighted_phase_lag_index", so.debiased_squared_weighted_phase_lag_index),
              ("phase_lag_index", so.phase_lag_index))
    for name, ref_fn in checks:
        c = Connectivity.from_multitaper(Multitaper(x, **kw), dtype=np.complex64)       # a fresh object: the first request picks the format
        got = getattr(c, name)()
        assert c._spectra.P is not None and c._spectra._X is None, name
        ref = ref_fn(coef)
        if name == "phase_lag_index":
            # a sign can flip where Im s is at the float32 rounding level: a few entries may differ by 2 / n_observations
            n = c.n_observations
            bad = np.abs(got - ref) > 1e-6
            assert bad.mean() < 2e-4 and np.all(np.abs(got - ref)[bad] <= 2.0 / n + 1e-6), name
        else:
            np.testing.assert_allclose(got, ref, rtol=5e-4, atol=5e-5, equal_nan=True, err_msg=name)


# ---- round 5: the three-term arithmetic, the split-bin parts and the dynamic range of the format, at depth -----------------------
def _planes_from_complex64(X, C):
    """Dense complex64 spectra [F][W][R][K][C] -> (planes buffer, scales) through the C ABI's conversion."""
    lib = _lib.load()
    F, W, R, K, _ = X.shape
    sp = engine.DeviceSpectra(X, (F, W, R, K, C), (W * R * K * C, R * K * C, K * C, C), 2 * (F - 1), True, C_alloc=C)
    P = torch.zeros((F * W * R * K * lib.sc_planes_row_bytes(C),), dtype=torch.uint8, device=X.device)
    scale = torch.empty((2 * C,), dtype=torch.float32, device=X.device)
    work = torch.empty((C,), dtype=torch.int32, device=X.device)
    _lib.check(lib.sc_planes_scales_from_spectra_f32(X.data_ptr(), F * W * R * K, C, scale.data_ptr(), work.data_ptr(), None), "scales")
    _lib.check(lib.sc_planes_from_spectra_f32(X.data_ptr(), byref(sp.desc("trials_tapers")), scale.data_ptr(), P.data_ptr(), None), "to planes")
    return engine.DeviceSpectra(None, (F, W, R, K, C), sp.strides, sp.n_fft, True, C_alloc=C, P=P, scale=scale)


@pytest.mark.parametrize("C", [44, 64, 96, 128, 130, 256])
@pytest.mark.parametrize("n_obs", [255, 256, 511, 512, 513, 1536, 7000])
def test_stage_b_on_planes_three_term_sweep(n_obs, C, debug_env):
    """fused2_kernel against float64 sums of the same coefficients at the depths where its arithmetic changes: 255 / 256
    observations per bin (four cross terms below 256, three from there on: sc_fused2.hip fused2_setup), the 512-observation fold
    interval and its neighbours, 1536, and the 7000 of BASELINE configs[2]; 44 ... 256 signals (one launch up to 128, the staircase
    launches above; 64-observation chunks up to 64 signals); every bin split over 1, 2, 3 and 5 workgroups (SC_FUSED_SPLIT),
    folded by the library (fold=True) and left as partial records for the epilogue (fold=False) -- which must agree bit for bit.
    Bound on every entry of the upper triangle, un-normalised sums: |err| <= 3e-6 |S_ij| + 2e-7 sqrt(P_i P_j) for the cross-
    spectra (the full-depth bound of tests/test_gpu_full_depth.py with the pair's own scale sqrt(P_i P_j) in place of the array
    maximum: channel amplitudes are spread over six decades here), 1e-5 relative for the positive sums of |Im s|."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import unpack_record_planes
    from fp64_device_ref import sums_fp64
    dev = _dev()
    F, W, R, K = 3, 1, n_obs, 1
    rng = np.random.default_rng(1000 * C + n_obs)
    coef = rng.standard_normal((F, W, R, K, C)) + 1j * rng.standard_normal((F, W, R, K, C))
    coef = coef + 0.4 * coef[..., :1]                                  # a shared component (before the scales: no cancelling pairs)
    coef = coef * (0.2 + rng.random(C)) * 10.0 ** rng.integers(-3, 4, C)
    X = torch.from_numpy(coef.astype(np.complex64)).to(dev)
    spp = _planes_from_complex64(X, C)
    csm, ab = sums_fp64(X.to(torch.complex128))                        # un-normalised float64 sums of the complex64-rounded input
    S, A = csm.cpu().numpy().reshape(F, C, C), ab.cpu().numpy().reshape(F, C, C)
    Pw = np.real(np.einsum("fii->fi", S))
    pair = np.sqrt(Pw[:, :, None] * Pw[:, None, :])
    upper = np.triu(np.ones((C, C), dtype=bool))
    strict = np.triu(np.ones((C, C), dtype=bool), 1)
    worst = {}
    variants = [(s, None) for s in (1, 2, 3, 5)]
    variants += [(1, 3)] if n_obs < 256 else [(1, 4)]                   # the other arithmetic at this depth, forced (SC_FUSED2_TERMS)
    for split, terms in variants:
        debug_env("SC_FUSED_SPLIT", split)
        debug_env("SC_FUSED2_TERMS", terms)
        accum, n = engine.accumulate(spp, "trials_tapers", PLANES)
        assert n == n_obs and accum.dim() == 2
        rec = unpack_record_planes(accum.cpu().numpy(), C)                                  # [F, 3, C, C]
        got_S = rec[:, 0] + 1j * rec[:, 1]
        err = np.abs(got_S - S)[:, upper] / (3e-6 * np.abs(S)[:, upper] + 2e-7 * pair[:, upper])
        rel_a = np.abs(rec[:, 2] - A)[:, strict] / A[:, strict]
        worst[(split, terms)] = (err.max(), rel_a.max())
        assert err.max() <= 1.0, (split, terms, err.max())
        assert rel_a.max() <= 1e-5, (split, terms, rel_a.max())
        parts, _ = engine.accumulate(spp, "trials_tapers", PLANES, fold=False)
        if parts.dim() == 3:
            assert C <= 128 and split > 1 and parts.shape[0] == split
            assert torch.equal(engine.fold_parts(parts), accum), "partial records summed in part order differ from the library's fold"
            for which in (_lib.M_WPLI, _lib.M_COHERENCE_MAGNITUDE):
                a = engine.measure(parts, C, PLANES, n, which)
                b = engine.measure(accum, C, PLANES, n, which)
                assert torch.equal(a.nan_to_num(), b.nan_to_num()), "the epilogue on partial records differs from the epilogue on their sum"
            two = engine.measure_multi(parts, C, PLANES, n, [_lib.M_COHERENCE_MAGNITUDE, _lib.M_WPLI])
            assert torch.equal(two[1], engine.measure(accum, C, PLANES, n, _lib.M_WPLI))
            for which in (_lib.M_POWER, _lib.M_CSM, _lib.M_COHERENCY):                        # sc_measure_parts: every measure
                for wide in (False, True):
                    a = engine.measure(parts, C, PLANES, n, which, wide=wide)
                    b = engine.measure(accum, C, PLANES, n, which, wide=wide)
                    ra, rb = (torch.view_as_real(v) if v.is_complex() else v for v in (a, b))
                    assert a.dtype == b.dtype and torch.equal(ra.nan_to_num(), rb.nan_to_num()), (which, wide)
        else:
            assert torch.equal(parts, accum)
    print(f"\n  n_obs {n_obs:5d}, {C:3d} signals: err / bound and |Im s| rel err per (split, forced terms): "
          + ", ".join(f"{k}: {v[0]:.2f} / {v[1]:.1e}" for k, v in worst.items()))


@pytest.mark.parametrize("artefact,offset", [(30.0, 0.0), (100.0, 0.0), (1e3, 0.0), (1e5, 0.0), (1e7, 0.0), (0.0, 3e3), (100.0, 3e3)])
def test_one_artefact_sample_or_a_dc_offset_in_a_channel(artefact, offset):
    """The format has ONE scale per channel.  (a) One sample 30 ... 1e7 times the channel's standard deviation (an electrode pop) in
    one trial: the quiet windows of that channel must keep their accuracy.  While the typical coefficient stays above
    _lib.PLANES_MIN_TYPICAL in scaled units (artefacts up to ~250 standard deviations at this window length) the format is kept and
    the measures of EVERY quiet window stay as close to the float64 oracle as the complex64 engine's; beyond, Multitaper says so
    (device_format_note, a logged warning) and the spectra are complex64 -- the same numbers as with SC_PLANES_FORMAT=0.
    (b) A DC offset thousands of times the signal (raw EEG / MEG): with a detrend active the scales come from the channel's RANGE, so
    the offset costs the format nothing (round 4 took the scale from max |x|: the coefficients then sat 3e3 times lower in scaled units)."""
    import os
    _dev()
    rng = np.random.default_rng(7)
    T, R, C, L, step = 1024, 6, 8, 256, 128
    t = np.arange(T) / 1000.0
    x = rng.standard_normal((T, R, C))
    x += 0.6 * np.sin(2 * np.pi * 60 * t[:, None, None] + 2 * np.pi * np.arange(C)[None, None, :] / C)
    if artefact:
        x[10, 0, 3] = artefact                      # window 0 of trial 0 only (samples 0 ... 255 with a 128-sample step)
    x[:, :, 5] += offset
    x = x.astype(np.float32)
    kw = dict(sampling_frequency=1000, time_halfbandwidth_product=3, n_time_samples_per_window=L, n_time_samples_per_step=step)
    m = Multitaper(x, **kw)
    c = Connectivity.from_multitaper(m, dtype=np.complex64)
    coh, wpli = c.coherence_magnitude(), c.weighted_phase_lag_index()
    kept = c._spectra.P is not None
    assert kept == (artefact <= 100.0), (artefact, offset, m.device_format_note)
    assert (m.device_format_note is None) == kept
    os.environ["SC_PLANES_FORMAT"] = "0"
    try:
        c64 = Connectivity.from_multitaper(Multitaper(x, **kw), dtype=np.complex64)
        coh64, wpli64 = c64.coherence_magnitude(), c64.weighted_phase_lag_index()
        assert c64._spectra.P is None
    finally:
        del os.environ["SC_PLANES_FORMAT"]
    coef, _ = so.multitaper_fft(x.astype(np.float64), fs=1000, NW=3, n_time_samples_per_window=L, n_time_samples_per_step=step)
    ref_coh, ref_wpli = so.coherence_magnitude(coef), so.weighted_phase_lag_index(coef)
    quiet = slice(1, None) if artefact else slice(None)          # the windows without the artefact
    e_p = np.nanmax(np.abs(coh - ref_coh)[quiet]), np.abs(wpli - ref_wpli)[quiet].max()
    e_c = np.nanmax(np.abs(coh64 - ref_coh)[quiet]), np.abs(wpli64 - ref_wpli)[quiet].max()
    print(f"\n  artefact x{artefact:g}, offset {offset:g}: planes kept {kept}" + (f" (typical coefficient {c._spectra.planes_typical_coefficient():.1f})" if kept else "")
          + f"; quiet windows vs float64 oracle: coherence {e_p[0]:.2e} (complex64 engine {e_c[0]:.2e}), wPLI {e_p[1]:.2e} ({e_c[1]:.2e})")
    # 30 observations per bin: what the float32 transform leaves of a measure is the same in both device formats; the format must not add to it
    assert e_p[0] <= 2 * e_c[0] + 2e-6 and e_p[1] <= 2 * e_c[1] + 2e-6
    if not offset:
        np.testing.assert_allclose(coh[quiet], ref_coh[quiet], rtol=2e-4, atol=5e-5, equal_nan=True)
        np.testing.assert_allclose(wpli[quiet], ref_wpli[quiet], rtol=2e-4, atol=5e-5)
    if not kept:
        np.testing.assert_array_equal(coh, coh64)
        np.testing.assert_array_equal(wpli, wpli64)
    if artefact:
        # the window that holds the artefact: the float32 transform itself is limited there (in either format); the two engines agree
        np.testing.assert_allclose(coh[0], coh64[0], rtol=0, atol=5e-5, equal_nan=True)


def test_prepare_keeps_the_planes_route_for_mixed_families():
    """wrapper.multitaper_connectivity([...]) prepares the accumulators of all its measures up front (Connectivity._prepare): the
    float32 engine makes one record per kernel family -- cross-spectral (+ |Im s|), sign(Im s) -- so that a list that mixes
    coherence, wPLI and PLI stays on the f16 pieces (a combined CSM + sign record exists in no planes kernel and would push the
    object back to complex64 spectra), and every measure then costs an epilogue."""
    _dev()
    x = _series(1024, 9, 12, seed=3, quiet=0.1, loud=8.0)
    m = Multitaper(x, sampling_frequency=1000, time_halfbandwidth_product=3, n_time_samples_per_window=256, n_time_samples_per_step=128)
    c = Connectivity.from_multitaper(m, dtype=np.complex64)
    names = ["coherence_magnitude", "phase_lag_index", "weighted_phase_lag_index"]
    c._prepare(names)
    assert c._spectra.P is not None and c._spectra._X is None
    assert sorted(k for k in c._accum_cache if isinstance(k, int)) == [PLANES], sorted(c._accum_cache)     # CSM + |Im s|; sign follows with PLI
    coef, _ = so.multitaper_fft(x, fs=1000, NW=3, n_time_samples_per_window=256, n_time_samples_per_step=128)
    got = {n: getattr(c, n)() for n in names}
    assert c._spectra._X is None, "a measure left the planes route"
    assert sorted(k for k in c._accum_cache if isinstance(k, int)) == [PLANES, _lib.PLANE_SIGN_IM]
    np.testing.assert_allclose(got["coherence_magnitude"], so.coherence_magnitude(coef), rtol=2e-4, atol=2e-5, equal_nan=True)
    np.testing.assert_allclose(got["weighted_phase_lag_index"], so.weighted_phase_lag_index(coef), rtol=2e-4, atol=2e-5)
    ref = so.phase_lag_index(coef)
    bad = np.abs(got["phase_lag_index"] - ref) > 1e-6
    assert bad.mean() < 2e-4 and np.all(np.abs(got["phase_lag_index"] - ref)[bad] <= 2.0 / c.n_observations + 1e-6)
