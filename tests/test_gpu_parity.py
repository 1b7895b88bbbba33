"""Parity of the HIP path (through the C ABI) against golden vectors from the real reference
and against the CPU oracle on seeded inputs.  Tolerance: the device pipeline computes in
fp32 / complex64 (fp64 only for trend sums and the measures epilogue); north-star bar is
1e-5 relative.  `close32` checks |a-b| <= rtol*|b| + atol_scale*max|b| -- i.e. 1e-5 relative
elementwise, with entries that are cancellation-small compared to the array's scale allowed
an absolute error of 1e-5 of that scale (they are sums of O(scale) terms in fp32)."""
import numpy as np
import pytest

from conftest import granger_close
from oracle import spectral_oracle as so

pytestmark = pytest.mark.gpu
SC_PRECISIONS = ("float32", "float32+planes", "dtype")     # under the forced float32 engine (complex64 spectra, and once more with the
# planes format of round  4 from two channels on) AND the package default (float64 engine for the
# default dtype): every test against the reference's golden vectors, the labelled wrapper, the dtype / NaN / complex-input
# behaviour and the eigen-solver.  The kernel-selection tests below exercise float32 kernels explicitly and run once
# (their float64 counterparts live in tests/test_gpu_fp64.py).
SC_PRECISIONS_TESTS = {
    "test_f1_cfg1", "test_f2_detrend", "test_f3_every_measure_every_expectation", "test_f4_lengths", "test_f7_edges",
    "test_known_answers_from_reference_unit_tests", "test_f5_granger_vs_reference", "test_granger_from_uploaded_two_sided_coefficients",
    "test_f12_cholesky_failure_outcome", "test_f6_canonical_coherence", "test_f9_mvar_measures_vs_reference",
    "test_f10_global_coherence_vs_reference", "test_wrapper_labelled_outputs_against_the_oracle",
    "test_f11_band_statistics_on_the_device_coherency", "test_output_dtypes_are_the_references",
    "test_nonfinite_sample_spoils_only_its_own_channel", "test_f13_canonical_coherence_with_fewer_observations_than_channels",
    "test_global_coherence_any_rank_beyond_64_signals", "test_global_coherence_degenerate_eigenvalues_and_the_jacobi_cross_check",
    "test_f14_complex_valued_time_series", "test_silent_and_constant_channels_give_exact_zero_spectra",
}

RTOL = 1e-5
ATOL_SCALE = 1e-5


def close32(a, b, rtol=RTOL, atol_scale=ATOL_SCALE, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} != {b.shape}"
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert np.array_equal(nan_a, nan_b), f"{what}: NaN pattern differs ({nan_a.sum()} vs {nan_b.sum()})"
    ok = ~nan_b
    if not ok.any():
        return
    scale = np.abs(b[ok]).max()
    err = np.abs(a[ok] - b[ok])
    bound = rtol * np.abs(b[ok]) + atol_scale * scale
    worst = (err / np.maximum(bound, 1e-300)).max()
    assert worst <= 1.0, (f"{what}: max err {err.max():.3e} (scale {scale:.3e}), "
                          f"worst err/bound {worst:.2f}")


@pytest.fixture(scope="module")
def sc():
    import spectral_connectivity_amd as pkg
    from spectral_connectivity_amd import _hosts, _lib
    if _hosts.kind() == "numpy":
        # the torch-free host (SC_HIP_HOST=numpy; tests/test_gpu_numpy_host_suite.py runs this module that way in a process of its
        # own): torch is never imported, conftest.pytest_sessionfinish checks that it was not
        _lib.require_gpu()
        return pkg
    import torch
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    _lib.load()
    return pkg


MEASURE_NAMES = list(so.MEASURES)


def test_f1_cfg1(sc, golden):
    g = golden("f1_cfg1")
    m = sc.Multitaper(g["x"], sampling_frequency=float(g["fs"]), time_halfbandwidth_product=float(g["NW"]))
    close32(m.fft(), g["fft"], what="fft")
    c = sc.Connectivity.from_multitaper(m)
    close32(c.power(), g["power"], what="power")
    close32(c.coherency(), g["coherency"], what="coherency")
    close32(c.coherence_magnitude(), g["coherence_magnitude"], what="coherence")
    np.testing.assert_allclose(c.frequencies, g["conn_frequencies"])


@pytest.mark.parametrize("det", ["constant", "linear", None])
def test_f2_detrend(sc, golden, det):
    g = golden("f2_detrend")
    m = sc.Multitaper(g["x"], sampling_frequency=float(g["fs"]), time_halfbandwidth_product=float(g["NW"]),
                      detrend_type=det)
    close32(m.fft(), g[f"fft_{det}"], what=f"fft detrend={det}")


@pytest.mark.parametrize("et", list(so.EXPECTATION_AXES))
def test_f3_every_measure_every_expectation(sc, golden, et):
    g = golden("f3_windows_all_measures")
    m = sc.Multitaper(g["x"], sampling_frequency=float(g["fs"]), time_halfbandwidth_product=float(g["NW"]),
                      n_time_samples_per_window=int(g["L"]), n_time_samples_per_step=int(g["step"]))
    c = sc.Connectivity.from_multitaper(m, expectation_type=et)
    for name in MEASURE_NAMES:
        ref = g[f"{et}__{name}"]
        got = getattr(c, name)()
        if name in ("phase_lag_index", "debiased_squared_phase_lag_index"):
            # sign(Im s) is discontinuous: an fp32 coefficient within rounding of Im s = 0 flips
            # one observation.  Require exactness up to a handful of flips.
            n = c.n_observations
            bad = np.abs(got - ref) > 1e-6 + 1e-5 * np.abs(ref)
            assert bad.mean() < 2e-3, f"{et}/{name}: {bad.sum()} of {bad.size} entries differ"
            assert np.nanmax(np.abs(got - ref)) <= 4.0 / n + 1e-6 if name == "phase_lag_index" else True
            continue
        if name == "coherence_phase":
            d = np.angle(np.exp(1j * (got - ref)))
            ok = ~np.isnan(ref)
            mag = g[f"{et}__coherence_magnitude"]
            # phase is ill-conditioned where coherence ~ 0: weight the error by |coherency|
            assert np.nanmax(np.abs(d[ok]) * np.sqrt(mag[ok])) < 2e-5, f"{et}/phase"
            continue
        if name == "debiased_squared_weighted_phase_lag_index":
            # (|sum Im|^2 - sum Im^2) / ((sum |Im|)^2 - sum Im^2) over as few as THREE observations: a ratio of
            # differences whose conditioning is unbounded.  Compared without any conditioning allowance: the three SUMS
            # the device accumulated, straight from its records, against the oracle's (each at the plain tolerance), and
            # the measure against the reference's formula evaluated on those same device sums (the fp64 epilogue: 1e-9).
            from conftest import device_record, unpack_record_planes
            from spectral_connectivity_amd import _lib
            planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM | _lib.PLANE_IM_SQ
            C = g["x"].shape[2]
            dev = unpack_record_planes(device_record(c, et, planes), C).reshape(ref.shape[:-3] + (ref.shape[-3], 4, C, C))
            d_im, d_abs, d_sq = (np.moveaxis(dev, -3, 0)[k] for k in (1, 2, 3))
            coef, _ = so.multitaper_fft(np.asarray(g["x"], dtype=np.float64), fs=float(g["fs"]), NW=float(g["NW"]),
                                        n_time_samples_per_window=int(g["L"]), n_time_samples_per_step=int(g["step"]))
            n = so.n_observations(coef, et)
            F = ref.shape[-3]
            o_im = (so.expectation_csm_faithful(coef, et, fcn=so._zero_diag_imag) * n)[..., :F, :, :]
            o_abs = (so.expectation_csm_faithful(coef, et, fcn=lambda s_: np.abs(so._zero_diag_imag(s_))) * n)[..., :F, :, :]
            o_sq = (so.expectation_csm_faithful(coef, et, fcn=lambda s_: so._zero_diag_imag(s_) ** 2) * n)[..., :F, :, :]
            iu = np.triu_indices(C, k=1)
            for what, a, b in (("sum Im s", d_im, o_im), ("sum |Im s|", d_abs, o_abs), ("sum (Im s)^2", d_sq, o_sq)):
                close32(a[..., iu[0], iu[1]], b[..., iu[0], iu[1]], what=f"{et}/{what}")
            w = d_abs ** 2 - d_sq
            with np.errstate(invalid="ignore", divide="ignore"):
                from_sums = np.where(w == 0, np.nan, (d_im ** 2 - d_sq) / w)
            up = got[..., iu[0], iu[1]]
            fs_ = from_sums[..., iu[0], iu[1]]
            assert np.array_equal(np.isnan(up), np.isnan(fs_)), f"{et}/{name}: NaN pattern vs the formula on the device sums"
            ok = ~np.isnan(fs_)
            tol_e = 1e-8            # same sums in, fp64 arithmetic on both sides: what is left is the order of operations
            assert np.abs(up[ok] - fs_[ok]).max() <= tol_e * max(np.abs(fs_[ok]).max(), 1.0), f"{et}/{name}: epilogue vs formula"
            np.testing.assert_allclose(np.swapaxes(got, -1, -2)[..., iu[0], iu[1]], up, rtol=0, atol=0, equal_nan=True)
            continue
        close32(got, ref, what=f"{et}/{name}")


@pytest.mark.parametrize("tag,kw", [
    ("L250", dict(n_time_samples_per_window=250)),
    ("L250_N300", dict(n_time_samples_per_window=250, n_fft_samples=300)),
    ("L255", dict(n_time_samples_per_window=255)),
    ("L256_N255", dict(n_time_samples_per_window=256, n_fft_samples=255)),
    ("dur_step", dict(time_window_duration=0.8, time_window_step=0.29)),
])
@pytest.mark.parametrize("kernel", ["engine's choice", "register passes"])
def test_f4_lengths(sc, golden, tag, kw, kernel, debug_env):
    # ("register passes": csrc/sc_mtfft_mixed.hip whatever the size -- 250 and 300 samples here; the engine by itself takes it from
    #  256 (window, trial, channel tile) items on and for every planes-format request)
    debug_env("SC_MTFFT_MIXED", "1" if kernel == "register passes" else None)
    g = golden("f4_lengths")
    m = sc.Multitaper(g["x"], sampling_frequency=float(g["fs"]), time_halfbandwidth_product=float(g["NW"]), **kw)
    close32(m.fft(), g[f"{tag}__fft"], what=f"{tag} fft")
    c = sc.Connectivity.from_multitaper(m)
    close32(c.coherence_magnitude(), g[f"{tag}__coherence_magnitude"], what=f"{tag} coherence")
    close32(c.power(), g[f"{tag}__power"], what=f"{tag} power")
    np.testing.assert_allclose(c.frequencies, g[f"{tag}__conn_frequencies"])


def test_f7_edges(sc, golden):
    g = golden("f7_edges")
    m = sc.Multitaper(g["zero__x"], sampling_frequency=100.0, time_halfbandwidth_product=2)
    c = sc.Connectivity.from_multitaper(m)
    for name in ("coherence_magnitude", "imaginary_coherence", "weighted_phase_lag_index"):
        close32(getattr(c, name)(), g[f"zero__{name}"], what=f"zero-power {name}")
    m = sc.Multitaper(g["nw175__x"], sampling_frequency=100.0, tapers=g["user__tapers"])
    close32(m.fft(), g["user__fft"], what="user tapers fft")
    # raw 5-D coefficients uploaded straight into Connectivity (reference tests build these)
    c = sc.Connectivity(g["raw__coef"], dtype=np.complex64)
    close32(c._expectation_cross_spectral_matrix(), g["raw__csm"][:, :9], what="raw csm")
    close32(c.coherence_magnitude(), g["raw__coherence_magnitude"], what="raw coherence")


def test_known_answers_from_reference_unit_tests(sc):
    # reference tests/test_connectivity.py:25-56, :82-99, :137-161, :200-264
    coef = np.zeros((1, 1, 1, 1, 2), dtype=complex)
    coef[..., 0] = 2 * np.exp(1j * np.pi / 2)
    coef[..., 1] = 3 * np.exp(-1j * np.pi / 2)
    c = sc.Connectivity(coef)
    close32(c._expectation_cross_spectral_matrix()[0, 0], np.array([[4, -6], [-6, 9]], complex), what="csm KAT")
    close32(c.power()[0, 0], np.array([4.0, 9.0]), what="power KAT")
    coh = c.coherency()[0, 0]
    assert np.isnan(coh[0, 0]) and np.isnan(coh[1, 1])
    assert abs(abs(coh[0, 1]) - 1) < 1e-6 and abs(abs(np.angle(coh[0, 1])) - np.pi) < 1e-6
    coef[..., 1] = 3.0
    c = sc.Connectivity(coef)
    np.testing.assert_allclose(c.phase_lag_index()[0, 0], [[0, 1], [-1, 0]], atol=1e-7)
    np.testing.assert_allclose(c.weighted_phase_lag_index()[0, 0], [[0, 1], [-1, 0]], atol=1e-6)
    np.testing.assert_allclose(c.phase_locking_value()[0, 0, 0, 1], 1.0, atol=1e-6)


@pytest.mark.parametrize("C,R,et", [(32, 20, "trials_tapers"), (128, 6, "trials_tapers"),
                                    (40, 5, "trials"), (19, 7, "time_trials_tapers"), (160, 3, "trials_tapers")])
def test_seeded_vs_oracle_multi_tile(sc, C, R, et):
    """Channel counts that exercise 1..10 channel blocks, odd sizes and the tile mirroring."""
    rng = np.random.default_rng(100 + C)
    T = 384
    x = rng.standard_normal((T, R, C))
    t = np.arange(T) / 500.0
    x += 0.6 * np.sin(2 * np.pi * 45 * t)[:, None, None] * rng.standard_normal(C)[None, None, :]
    x += 0.6 * np.cos(2 * np.pi * 45 * t)[:, None, None] * rng.standard_normal(C)[None, None, :]
    kw = dict(n_time_samples_per_window=128, n_time_samples_per_step=64)
    m = sc.Multitaper(x, sampling_frequency=500.0, time_halfbandwidth_product=2, **kw)
    c = sc.Connectivity.from_multitaper(m, expectation_type=et)
    coef, _ = so.multitaper_fft(x, fs=500.0, NW=2, **kw)
    csm = so.expectation_csm_gemm(coef, et)
    close32(c.coherence_magnitude(), so.coherence_magnitude(coef, et, csm=csm), what="coherence")
    close32(c.imaginary_coherence(), so.imaginary_coherence(coef, et, csm=csm), what="imag coh")
    close32(c.power(), so.power(coef, et), what="power")
    if C <= 40:
        close32(c.weighted_phase_lag_index(), so.weighted_phase_lag_index(coef, et), what="wpli")
        # PLV normalises every observation to unit modulus, so the fp32 phase error of the
        # SMALLEST coefficients enters at full weight: 3e-5 is the honest fp32 bar here
        close32(c.phase_locking_value(), so.phase_locking_value(coef, et), rtol=3e-5, atol_scale=3e-5,
                what="plv")
        close32(c.debiased_squared_weighted_phase_lag_index(),
                so.debiased_squared_weighted_phase_lag_index(coef, et), rtol=1e-4, atol_scale=1e-4,
                what="dwpli2")


@pytest.mark.parametrize("N,L,C,det", [(64, 64, 5, "constant"), (128, 100, 70, "linear"), (256, 256, 33, None),
                                       (512, 300, 16, "constant"), (1024, 1024, 18, "linear"),
                                       (2048, 2048, 6, "constant"), (4096, 4000, 3, "constant"),
                                       (2048, 1500, 5, "linear"), (4096, 4096, 7, None), (2048, 2048, 130, "constant"),
                                       (4096, 3000, 1, "linear"), (1024, 700, 11, "constant")])
def test_fused_fft_matches_oracle_and_rocfft(sc, N, L, C, det):
    """Fused HIP transform (power-of-two N, zero padding, odd channel counts, every detrend)
    against the float64 oracle, and the rocFFT fallback path against the same oracle."""
    import torch
    from spectral_connectivity_amd import engine
    rng = np.random.default_rng(N + C)
    R, step = 3, max(L // 2, 1)
    T = L + 2 * step
    x = rng.standard_normal((T, R, C)) + 5.0 + np.linspace(0, 2, T)[:, None, None]
    kw = dict(n_time_samples_per_window=L, n_time_samples_per_step=step, n_fft_samples=N)
    coef, info = so.multitaper_fft(x, fs=200.0, NW=2.5, detrend_type=det, **kw)
    m = sc.Multitaper(x, sampling_frequency=200.0, time_halfbandwidth_product=2.5, detrend_type=det, **kw)
    close32(m.fft(), coef, what=f"fused N={N}")
    xd = torch.from_numpy(x.astype(np.float32)).cuda()
    h = torch.from_numpy(np.ascontiguousarray(m.tapers.T / 200.0, dtype=np.float32)).cuda()
    sp = engine.multitaper_spectra(xd, h, L, step, N, m.n_time_windows, det, use_fused=False)
    got = np.moveaxis(sp.coefficients().cpu().numpy(), 0, 3)
    close32(got, coef[..., : N // 2 + 1, :], what=f"rocfft N={N}")


@pytest.mark.parametrize("C,R", [(128, 9), (96, 5), (64, 6), (24, 11), (6, 4), (128, 40), (2, 50), (16, 300), (32, 7),
                                 (34, 5), (42, 9), (44, 4), (48, 6), (50, 3), (52, 4), (56, 3), (130, 4), (160, 6), (162, 3), (176, 3),
                                 (192, 3), (194, 3), (208, 4), (224, 3), (226, 2), (250, 3), (256, 5)])
def test_fused_stage_b_equals_separate_kernels(sc, C, R):
    """The one-pass stage-B kernels -- bf16 MFMA + VALU from 50 channels on (44 without the |Im| plane), f32 VALU
    below -- against the separate f32-MFMA CSM and |Im| kernels on the same spectra (identical fp32 arithmetic per
    plane up to summation order)."""
    import torch
    from spectral_connectivity_amd import _lib, engine
    rng = np.random.default_rng(C)
    x = rng.standard_normal((300, R, C))
    m = sc.Multitaper(x, sampling_frequency=300.0, time_halfbandwidth_product=3,
                      n_time_samples_per_window=128, n_time_samples_per_step=64)
    sp = m.device_spectra()
    planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
    a_f, n = engine.accumulate(sp, "trials_tapers", planes, use_fused=True)
    a_s, _ = engine.accumulate(sp, "trials_tapers", planes, use_fused=False)
    for which in (_lib.M_CSM, _lib.M_WPLI, _lib.M_COHERENCE_MAGNITUDE):
        got = engine.measure(a_f, C, planes, n, which).cpu().numpy()
        ref = engine.measure(a_s, C, planes, n, which).cpu().numpy()
        close32(got, ref, rtol=2e-6, atol_scale=2e-6, what=f"fused vs separate, measure {which}")
    # CSM alone through the fused kernel (matrix-core role only)
    c_f, _ = engine.accumulate(sp, "trials_tapers", _lib.PLANE_CSM, use_fused=True)
    c_s, _ = engine.accumulate(sp, "trials_tapers", _lib.PLANE_CSM, use_fused=False)
    close32(engine.measure(c_f, C, _lib.PLANE_CSM, n, _lib.M_CSM).cpu().numpy(),
            engine.measure(c_s, C, _lib.PLANE_CSM, n, _lib.M_CSM).cpu().numpy(), rtol=2e-6, atol_scale=2e-6,
            what="fused CSM only vs f32 MFMA kernel")
    coef, _ = so.multitaper_fft(x, fs=300.0, NW=3, n_time_samples_per_window=128, n_time_samples_per_step=64)
    if C <= 64:
        close32(sc.Connectivity.from_multitaper(m).weighted_phase_lag_index(),
                so.weighted_phase_lag_index(coef), what="wpli vs oracle")


@pytest.mark.parametrize("C,R", [(2, 40), (6, 9), (16, 120), (34, 5), (40, 6), (42, 5), (48, 7), (50, 4), (52, 5), (54, 4), (58, 6), (60, 3), (64, 6), (96, 4), (128, 5),
                                 (129, 3), (130, 3), (160, 4), (162, 2), (192, 3), (194, 2), (224, 2), (226, 2), (255, 2), (256, 2)])
def test_one_pass_nonlinear_planes_equal_the_per_plane_kernel(sc, C, R):
    """(Im s)^2 and sign(Im s) ride on the small-channel one-pass kernel (<= 58 / <= 44 channels) or are plane passes of the
    matrix-core kernel (up to 128), and the unit phasors s/|s| go through the one-pass kernels as the cross-spectral
    matrix of x/|x| at every size: the same accumulator records as the
    per-plane VALU kernel (sc_nonlinear.hip) up to f32 summation order -- sign sums exactly."""
    from spectral_connectivity_amd import _lib, engine
    rng = np.random.default_rng(100 + C)
    x = rng.standard_normal((200, R, C)) + 0.5 * rng.standard_normal((200, R, 1))
    m = sc.Multitaper(x, sampling_frequency=200.0, time_halfbandwidth_product=3,
                      n_time_samples_per_window=64, n_time_samples_per_step=32)
    sp = m.device_spectra()
    for planes, which in ((_lib.PLANE_CSM | _lib.PLANE_ABS_IM | _lib.PLANE_IM_SQ, _lib.M_DEBIASED_WPLI2),
                          (_lib.PLANE_SIGN_IM, _lib.M_PLI), (_lib.PLANE_SIGN_IM, _lib.M_DEBIASED_PLI2),
                          (_lib.PLANE_UNIT, _lib.M_PLV), (_lib.PLANE_UNIT, _lib.M_PPC)):
        for et in ("trials_tapers", "time_trials_tapers"):
            a_f, n = engine.accumulate(sp, et, planes, use_fused=True)
            a_s, _ = engine.accumulate(sp, et, planes, use_fused=False)
            got = engine.measure(a_f, C, planes, n, which).cpu().numpy()
            ref = engine.measure(a_s, C, planes, n, which).cpu().numpy()
            if planes == _lib.PLANE_SIGN_IM:
                assert np.array_equal(np.isnan(got), np.isnan(ref))
                if C <= 44:                  # both kernels form Im s with the same f32 operations: identical sign sums
                    assert np.array_equal(got[~np.isnan(got)], ref[~np.isnan(ref)])
                else:
                    # 42 ... 256 channels: the signs come from the matrix-core products (six bf16 cross terms, f32
                    # sums) instead of an f32 FMA pair -- an observation whose |Im s| is within f32 rounding of zero
                    # may land on the other side: a handful of entries off by one or two flipped observations
                    diff = np.abs(got - ref)[~np.isnan(ref)]
                    assert (diff > 0).mean() < 1e-3 and diff.max() <= 8.0 / n, (C, et, (diff > 0).sum(), diff.max())
            else:
                tol = 2e-5 if which == _lib.M_DEBIASED_WPLI2 else 3e-6     # the debiased ratio amplifies re-association
                close32(got, ref, rtol=tol, atol_scale=tol, what=f"planes {planes:#x} measure {which} {et}")


@pytest.mark.parametrize("C,R,split", [(128, 150, 3), (64, 130, 2), (128, 200, 5), (16, 400, 4), (40, 130, 3), (4, 900, 7)])
def test_fused_stage_b_split_bins(sc, C, R, split, debug_env):
    """Several workgroups per bin (observation chunks split, partial records folded in a fixed order)
    give the sums of the one-workgroup-per-bin launch up to fp32 re-association, and repeat bit-exactly."""
    from spectral_connectivity_amd import _lib, engine
    rng = np.random.default_rng(C + R)
    x = rng.standard_normal((128, R, C))
    m = sc.Multitaper(x, sampling_frequency=128.0, time_halfbandwidth_product=3,
                      n_time_samples_per_window=64, n_time_samples_per_step=64)
    sp = m.device_spectra()
    planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
    debug_env("SC_FUSED_SPLIT", "1")
    a_1, n = engine.accumulate(sp, "trials_tapers", planes, use_fused=True)
    debug_env("SC_FUSED_SPLIT", str(split))
    a_s, _ = engine.accumulate(sp, "trials_tapers", planes, use_fused=True)
    a_s2, _ = engine.accumulate(sp, "trials_tapers", planes, use_fused=True)
    assert bool((a_s == a_s2).all()), "split launch is not reproducible"
    for which in (_lib.M_CSM, _lib.M_WPLI, _lib.M_COHERENCE_MAGNITUDE):
        got = engine.measure(a_s, C, planes, n, which).cpu().numpy()
        ref = engine.measure(a_1, C, planes, n, which).cpu().numpy()
        close32(got, ref, rtol=2e-6, atol_scale=2e-6, what=f"split {split} vs 1, measure {which}")
    coef, _ = so.multitaper_fft(x, fs=128.0, NW=3, n_time_samples_per_window=64, n_time_samples_per_step=64)
    got = engine.measure(a_s, C, planes, n, _lib.M_COHERENCE_MAGNITUDE).cpu().numpy()
    ref = so.coherence_magnitude(coef)
    close32(got.reshape(ref.shape), ref, what="split coherence vs oracle")


@pytest.mark.parametrize("tag,kw", [
    ("ding2", dict(time_halfbandwidth_product=1)),
    ("bacc3", dict(time_halfbandwidth_product=2, n_time_samples_per_window=250)),
])
def test_f5_granger_vs_reference(sc, golden, tag, kw):
    """Pairwise spectral Granger through the batched 2x2 Wilson kernel vs the real reference."""
    g = golden("f5_granger")
    m = sc.Multitaper(g[f"{tag}__x"], sampling_frequency=200.0, **kw)
    c = sc.Connectivity.from_multitaper(m)
    got = c.pairwise_spectral_granger_prediction()
    ref = g[f"{tag}__granger"]
    # values are log-ratios built from an fp32 CSM; NaN pattern (non-positive values) can flip
    # for entries that are ~0: compare where both are finite and require few flips
    granger_close(got, ref, 2e-5, what=tag)
    assert c._last_wilson["not_converged"] == 0
    # subset == full on the requested pairs (reference tests/test_connectivity.py:591-613)
    sub = c.subset_pairwise_spectral_granger_prediction([(0, 1)])
    np.testing.assert_allclose(sub[..., 0, 1], got[..., 0, 1], rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(sub[..., 1, 0], got[..., 1, 0], rtol=1e-12, equal_nan=True)


def test_granger_from_uploaded_two_sided_coefficients(sc, golden):
    g = golden("f5_granger")
    coef, _ = so.multitaper_fft(g["ding2__x"], fs=200.0, NW=1)
    c = sc.Connectivity(coef)
    got = c.pairwise_spectral_granger_prediction()
    ref = g["ding2__granger"]
    granger_close(got, ref, 2e-5, what="uploaded two-sided coefficients")


def test_f12_cholesky_failure_outcome(sc, golden):
    """A window whose lag-0 covariance has no Cholesky factor (a channel silent in the second window).  The reference
    restarts every window of the affected pairs from a random positive-definite matrix (minimum_phase_decomposition.py:
    78-93), this engine starts the failing problems from the identity: what is compared is the OUTCOME -- the good window
    converges to the same prediction from either start (the reference's own seed-to-seed spread is 1.5e-5: both of its
    runs are in the fixture), pairs that do not touch the silent channel are untouched, and the degenerate window's
    predictions for the silent channel carry no information on either side (NaN, or noise below 1e-9)."""
    g = golden("f12_cholesky_fallback")
    m = sc.Multitaper(g["x"], sampling_frequency=float(g["fs"]), time_halfbandwidth_product=float(g["NW"]),
                      n_time_samples_per_window=int(g["L"]))
    c = sc.Connectivity.from_multitaper(m)
    got = c.pairwise_spectral_granger_prediction()
    ref0, ref1 = g["granger_seed0"], g["granger_seed1"]
    assert got.shape == ref0.shape == (2, 129, 3, 3)
    assert c._last_wilson["cholesky_fallbacks"] >= 1
    spread = np.nanmax(np.abs(ref0[0] - ref1[0])) / np.nanmax(ref0[0])
    granger_close(got[0], ref0[0], max(4 * spread, 2e-5), what="good window, every pair")
    granger_close(got[1][:, :2, :2], ref0[1][:, :2, :2], 2e-5, what="degenerate window, pair without the silent channel")
    for i, j in ((0, 2), (2, 0), (1, 2), (2, 1)):
        for side in (got[1][:, i, j], ref0[1][:, i, j], ref1[1][:, i, j]):
            assert np.all(np.isnan(side) | (np.abs(side) < 1e-9)), (i, j)


@pytest.mark.parametrize("N,W", [(256, 3), (512, 2), (1024, 2), (2048, 1), (4096, 1),
                                 # (round 6) windows that are not powers of two: wilson_pair_mixed_kernel, N = P M -- every P, every radix
                                 (250, 3), (500, 2), (300, 2), (200, 3), (400, 2), (600, 2), (1000, 2), (800, 1), (1200, 1), (1600, 1),
                                 (2000, 1), (2400, 1), (3200, 1), (4000, 1)])
def test_granger_resident_kernel_equals_the_batched_kernels(sc, debug_env, N, W):
    """Pairwise spectral Granger of real series with a power-of-two window of 256 ... 4096 samples (round 6: and fourteen other lengths
    200 ... 4000, a mixed-radix transform in the same kernel) runs the whole 2 x 2 Wilson
    iteration of a pair on one compute unit (sc_wilson_pair.hip: half spectra in registers, two packed transforms per direction,
    convergence tested in the kernel); SC_GRANGER_KERNEL=batched keeps the three-kernels-per-iteration form of sc_wilson.hip
    (reference minimum_phase_decomposition.py:227-322 statement by statement).  Same records in: the predictions agree to the
    rounding of the transforms, every problem takes the same number of iterations and reaches the same status -- also when the
    iteration limit cuts the problems short."""
    import torch
    from spectral_connectivity_amd import _lib, engine
    rng = np.random.default_rng(N + W)
    C, R = 5, 6
    T = N * W
    e = rng.standard_normal((T + 64, R, C))
    x = np.zeros_like(e)
    for t in range(2, T + 64):
        x[t] = 0.5 * x[t - 1] - 0.3 * x[t - 2] + e[t]
        x[t, :, 1:] += 0.35 * x[t - 1, :, :-1]
        x[t, :, 0] += 0.2 * x[t - 2, :, 3]
    x = x[64:]
    m = sc.Multitaper(x, sampling_frequency=500.0, time_halfbandwidth_product=3, n_time_samples_per_window=N)
    c = sc.Connectivity.from_multitaper(m)
    pairs = np.array([(i, j) for i in range(C) for j in range(i + 1, C)], dtype=np.int32)
    accum, n_obs, n_freq = c._csm_records("granger")
    assert n_freq == N // 2 + 1 and accum.shape[0] == W * n_freq
    out = {}
    for kernel in ("batched", None):
        debug_env("SC_GRANGER_KERNEL", kernel)
        for max_it in (60, 3):
            gp, n_iter, status, summary = engine.granger_pairwise(accum, W, n_freq, N, C, _lib.PLANE_CSM, n_obs, pairs, max_iterations=max_it)
            out[(kernel, max_it)] = (gp.cpu().numpy(), n_iter.cpu().numpy(), status.cpu().numpy(), summary)
    for max_it in (60, 3):
        (gb, ib, sb, sumb), (gr, ir, sr, sumr) = out[("batched", max_it)], out[(None, max_it)]
        assert np.array_equal(ib, ir) and np.array_equal(sb, sr) and tuple(sumb) == tuple(sumr), (max_it, sumb, sumr)
        assert np.array_equal(np.isnan(gb), np.isnan(gr))
        ok = ~np.isnan(gb)
        assert np.abs(gb[ok] - gr[ok]).max() <= 1e-9 * np.abs(gb[ok]).max(), (max_it, np.abs(gb[ok] - gr[ok]).max())
    assert out[(None, 60)][3][1] == 0 and out[(None, 3)][3][1] == len(pairs) * W          # all converged / none within three iterations
    ref = so.pairwise_spectral_granger_prediction(so.multitaper_fft(x, fs=500.0, NW=3, n_time_samples_per_window=N)[0]) if N <= 512 else None
    if ref is not None:
        granger_close(out[(None, 60)][0].reshape(ref.shape), ref, 2e-5 if c._precision == "float32" else 1e-8, what="resident kernel vs the oracle")


def test_f6_canonical_coherence(sc, golden):
    g = golden("f6_canonical")
    m = sc.Multitaper(g["x"], sampling_frequency=float(g["fs"]), time_halfbandwidth_product=float(g["NW"]),
                      n_time_samples_per_window=int(g["L"]))
    c = sc.Connectivity.from_multitaper(m)
    cc, labels = c.canonical_coherence(g["group_labels"])
    assert np.array_equal(labels, g["labels"])
    close32(cc, g["canonical_coherence"], rtol=2e-5, atol_scale=2e-5, what="canonical coherence")


def test_minimum_phase_decomposition_vs_reference_and_known_filters(sc, golden):
    """Standalone Wilson factorisation: golden 2x2 factors from the real reference, and the
    reference's own known-answer idea (tests/test_minimum_phase_decomposition.py:96-119): a
    minimum-phase FIR filter is recovered from its power spectrum."""
    from spectral_connectivity_amd.minimum_phase_decomposition import minimum_phase_decomposition
    g = golden("f5_granger")
    for tag in ("ding2", "bacc3"):
        csm = g[f"{tag}__csm"][..., :2, :2]
        G = minimum_phase_decomposition(csm)
        np.testing.assert_allclose(G, g[f"{tag}__wilson01"], rtol=1e-6, atol=1e-8)
    N = 128
    w = 2 * np.pi * np.fft.fftfreq(N)
    H = 1.5 * (1 - 0.5 * np.exp(-1j * w)) * (1 + 0.3 * np.exp(-1j * w))        # zeros inside the unit circle
    S = (np.abs(H) ** 2)[None, :, None, None].astype(complex)
    G = minimum_phase_decomposition(S)
    np.testing.assert_allclose(G[0, :, 0, 0], H, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(G * np.conj(G), S, rtol=1e-7, atol=1e-9)        # exact for a rational spectrum
    G = minimum_phase_decomposition(np.tile(np.eye(3, dtype=complex), (1, 8, 1, 1)))   # white spectrum: G = I
    np.testing.assert_allclose(G, np.tile(np.eye(3, dtype=complex), (1, 8, 1, 1)), atol=1e-12)
    G = minimum_phase_decomposition(np.tile(np.eye(129, dtype=complex), (1, 4, 1, 1)))   # (round 3: up to 256 signals)
    np.testing.assert_allclose(G, np.tile(np.eye(129, dtype=complex), (1, 4, 1, 1)), atol=1e-12)
    G = minimum_phase_decomposition(np.tile(np.eye(257, dtype=complex), (1, 4, 1, 1)))   # (round 6: up to 512)
    np.testing.assert_allclose(G, np.tile(np.eye(257, dtype=complex), (1, 4, 1, 1)), atol=1e-12)
    with pytest.raises(NotImplementedError):
        minimum_phase_decomposition(np.tile(np.eye(513, dtype=complex), (1, 4, 1, 1)))


@pytest.mark.parametrize("tag", ["var3", "var5"])
def test_f9_mvar_measures_vs_reference(sc, golden, tag):
    """Full C x C Wilson factor and the directed MVAR measures (DTF, DC, PDC, gPDC, dDTF) against the
    real reference's golden vectors and the oracle.  The spectra are fp32, the factorisation fp64:
    the tolerance is the fp32 input tolerance amplified by the conditioning of the factorisation."""
    g = golden("f9_mvar")
    x = g[f"{tag}__x"]
    m = sc.Multitaper(x, sampling_frequency=128.0, time_halfbandwidth_product=2, n_time_samples_per_window=256)
    c = sc.Connectivity.from_multitaper(m)
    G = c._minimum_phase_factor
    ref = g[f"{tag}__minimum_phase_factor"]
    assert G.shape == ref.shape
    # (G G^H only approximates an ESTIMATED csm -- the reference's factor has the same residual -- so the
    # factor itself is compared; exact reconstruction is asserted on rational spectra below)
    close32(G, ref, rtol=1e-4, atol_scale=1e-4, what="minimum phase factor")
    close32(c._noise_covariance, g[f"{tag}__noise_covariance"], rtol=1e-4, atol_scale=1e-4, what="noise covariance")
    close32(c._transfer_function, g[f"{tag}__transfer_function"], rtol=1e-4, atol_scale=1e-4, what="transfer function")
    close32(c._MVAR_Fourier_coefficients, g[f"{tag}__mvar_coefficients"], rtol=2e-4, atol_scale=2e-4, what="MVAR coef")
    for name in ("directed_transfer_function", "directed_coherence", "partial_directed_coherence",
                 "generalized_partial_directed_coherence", "direct_directed_transfer_function"):
        close32(getattr(c, name)(), g[f"{tag}__{name}"], rtol=2e-4, atol_scale=2e-4, what=name)
    assert c._last_wilson["not_converged"] == 0


@pytest.mark.parametrize("c,N,P", [(3, 64, 2), (8, 128, 3), (17, 64, 1), (40, 32, 2), (64, 32, 1),
                                   (65, 32, 2), (80, 48, 1), (96, 64, 1), (97, 32, 1), (128, 32, 2), (100, 256, 1),
                                   (129, 32, 1), (160, 32, 1), (200, 256, 1), (250, 48, 1), (256, 32, 1),
                                   (257, 32, 1), (306, 32, 2), (512, 32, 1)])
def test_full_wilson_factor_standalone_fp64(sc, c, N, P):
    """minimum_phase_decomposition() for c > 2 on exactly representable fp64 spectra of known
    minimum-phase filters: S = F F^H with F(z) = I + B z^-1 (||B|| < 1) factors back to F Q with the
    lag-0 normalisation of the reference, i.e. G G^H = S to fp64 accuracy and equals the oracle's G."""
    from spectral_connectivity_amd.minimum_phase_decomposition import minimum_phase_decomposition
    rng = np.random.default_rng(c * N)
    S = np.empty((P, N, c, c), dtype=np.complex128)
    z = np.exp(-2j * np.pi * np.arange(N) / N)
    for p in range(P):
        B = rng.standard_normal((c, c))
        B *= 0.5 / np.linalg.norm(B, 2)
        L = np.linalg.cholesky(np.eye(c) + 0.3 * np.ones((c, c)) / c)
        Fz = (np.eye(c)[None] + B[None] * z[:, None, None]) @ L
        S[p] = Fz @ np.conj(np.swapaxes(Fz, -1, -2))
    G = minimum_phase_decomposition(S)
    assert G.shape == S.shape and np.isfinite(G).all()
    np.testing.assert_allclose(G @ np.conj(np.swapaxes(G, -1, -2)), S, rtol=0, atol=1e-7 * np.abs(S).max())
    # (beyond 64 signals: explicit inverse + matrix-core products; beyond 128: panel-blocked inverse in global memory and products
    #  cut into 128 x 128 blocks -- up to 512 signals since round 6, panels of eight columns beyond 256)
    # (the oracle's numpy iteration takes a minute at 256 signals: the largest sizes are held to G G^H = S only)
    if c <= 17 or (c, N) in ((65, 32), (128, 32), (129, 32), (160, 32)):
        np.testing.assert_allclose(G, so.minimum_phase_decomposition(S), rtol=0, atol=1e-6 * np.abs(G).max())


@pytest.mark.parametrize("C", [72, 128, 130, 160, 306])
def test_mvar_measures_beyond_64_signals_vs_oracle(sc, C):
    """65 ... 128 signals: Wilson factor, transfer function, noise covariance, MVAR coefficients and the directed
    measures through the explicit-inverse / matrix-core kernels (sc_mvar.hip), float64 engine, against the oracle;
    129 ... 512 signals: the same iteration on the panel-blocked inverse and the blocked products (306: a whole-head MEG array,
    its record assembled from channel-block pairs by engine._accumulate_blocked)."""
    rng = np.random.default_rng(C)
    # (beyond 128 signals: 32 bins, the oracle's time; 306 signals: 16 bins, 204 trials x 3 tapers = 612 observations)
    T, R = (64 if C <= 128 else (32 if C <= 256 else 16)), (90 if C <= 128 else (C if C <= 256 else 204))
    e = rng.standard_normal((T + 8, R, C))
    x = e.copy()
    for t in range(2, T + 8):                                 # a sparse stable VAR(2): neighbours drive each other
        x[t] += 0.35 * x[t - 1] - 0.2 * x[t - 2]
        x[t, :, 1:] += 0.25 * x[t - 1, :, :-1]
    x = x[8:]
    kw = dict(sampling_frequency=128.0, time_halfbandwidth_product=2)
    c = sc.Connectivity.from_multitaper(sc.Multitaper(x, **kw), dtype=np.complex128)
    coef, _ = so.multitaper_fft(x, fs=128.0, NW=2)
    q = so.mvar_quantities(coef)

    def close(a, b, what, tol=1e-6):
        a, b = np.asarray(a), np.asarray(b)
        assert a.shape == b.shape, (what, a.shape, b.shape)
        err = np.abs(a - b).max() / np.abs(b).max()
        assert err < tol, f"{what}: {err:.2e}"
    # (the iteration stops on max |dG| < 1e-8: device and oracle may stop one iteration apart, a few 1e-7 of the factor)
    loose = 1e-6 if C <= 128 else 3e-6
    close(c._minimum_phase_factor, q["G"], "minimum phase factor", tol=loose)
    close(c._noise_covariance, q["noise_covariance"], "noise covariance", tol=loose)
    close(c._transfer_function, q["H"], "transfer function", tol=loose)
    close(c._MVAR_Fourier_coefficients, q["A"], "MVAR coefficients", tol=1e-5)
    close(c.directed_transfer_function(), so.directed_transfer_function(coef, q=q), "DTF", tol=loose)
    if C <= 256:                # (306 signals: the oracle's partial coherence alone takes half a minute; the kernels are those of 160)
        close(c.partial_directed_coherence(), so.partial_directed_coherence(coef, q=q), "PDC", tol=1e-5)
        close(c.direct_directed_transfer_function(), so.direct_directed_transfer_function(coef, q=q), "dDTF", tol=1e-5)
    assert c._last_wilson["not_converged"] == 0 and c._last_wilson["iterations"] < 60


@pytest.mark.parametrize("c,N,P", [(2, 256, 5), (2, 512, 3), (2, 1024, 3), (2, 2048, 2), (2, 4096, 2),
                                   (3, 256, 2), (4, 512, 1), (3, 2048, 1), (5, 4096, 1)])
def test_wilson_fused_causal_fft_vs_library_path(sc, debug_env, c, N, P):
    """Lengths 256..4096 run ifft -> causal mask -> fft as one fp64 LDS kernel (csrc/sc_wilson_fft.hip); every
    other length, and SC_WILSON_FFT=rocfft, takes rocFFT + a pointwise kernel.  Both must produce the same
    factor (fp64 rounding apart), reconstruct S, and match the oracle where it is quick."""
    from spectral_connectivity_amd.minimum_phase_decomposition import minimum_phase_decomposition
    rng = np.random.default_rng(c * N + P)
    S = np.empty((P, N, c, c), dtype=np.complex128)
    z = np.exp(-2j * np.pi * np.arange(N) / N)
    for p in range(P):
        B1 = rng.standard_normal((c, c)); B1 *= 0.6 / np.linalg.norm(B1, 2)
        B2 = rng.standard_normal((c, c)); B2 *= 0.25 / np.linalg.norm(B2, 2)
        L = np.linalg.cholesky(np.eye(c) + 0.3 * np.ones((c, c)) / c)
        Fz = (np.eye(c)[None] + B1[None] * z[:, None, None] + B2[None] * (z ** 2)[:, None, None]) @ L
        S[p] = Fz @ np.conj(np.swapaxes(Fz, -1, -2))
    debug_env("SC_WILSON_FFT", None)
    G = minimum_phase_decomposition(S)
    debug_env("SC_WILSON_FFT", "rocfft")
    G_lib = minimum_phase_decomposition(S)
    debug_env("SC_WILSON_FFT", None)
    assert G.shape == S.shape and np.isfinite(G).all()
    scale = np.abs(G).max()
    np.testing.assert_allclose(G, G_lib, rtol=0, atol=1e-10 * scale)
    np.testing.assert_allclose(G @ np.conj(np.swapaxes(G, -1, -2)), S, rtol=0, atol=1e-7 * np.abs(S).max())
    if N <= 512:
        np.testing.assert_allclose(G, so.minimum_phase_decomposition(S), rtol=0, atol=1e-6 * scale)


@pytest.mark.parametrize("rank", [1, 2, 4, 5])
def test_f10_global_coherence_vs_reference(sc, golden, rank):
    """Leading eigenpairs of the device CSM against the reference's SVD of the coefficient matrix."""
    g = golden("f10_global")
    m = sc.Multitaper(g["x"], sampling_frequency=256.0, time_halfbandwidth_product=2, n_time_samples_per_window=128)
    c = sc.Connectivity.from_multitaper(m)
    vals, vecs = c.global_coherence(max_rank=rank)
    close32(vals, g[f"rank{rank}__values"], rtol=2e-5, atol_scale=2e-6, what=f"global coherence rank {rank}")
    ref = g[f"rank{rank}__vectors"]
    assert vecs.shape == ref.shape
    np.testing.assert_allclose(np.linalg.norm(vecs, axis=-2), 1.0, atol=1e-10)
    # same lines wherever the eigenvalue is separated from its neighbours (relative gap > 1e-2)
    allv = g["rank5__values"]                                   # descending, every eigenvalue
    ip = np.abs(np.sum(np.conj(vecs) * ref, axis=-2))
    order = (np.arange(rank)[::-1] if rank < 4 else np.arange(rank))          # position in the descending list
    for k in range(rank):
        idx = order[k]
        lam = allv[..., idx]
        gap = np.minimum(np.abs(allv[..., max(idx - 1, 0)] - lam) if idx > 0 else np.inf,
                         np.abs(allv[..., min(idx + 1, 4)] - lam) if idx < 4 else np.inf)
        sep = gap > 1e-2 * allv[..., 0]
        assert sep.mean() > 0.5
        assert (ip[..., k][sep] > 1 - 1e-3).all(), f"vector {k}: min |<u, u_ref>| = {ip[..., k][sep].min()}"


def test_global_coherence_large_even_and_odd(sc):
    """33, 64 (matrix + eigenvectors in LDS) and 65, 97, 128 signals (packed matrix in LDS, eigenvectors from the
    logged rotations) against the oracle SVD: values, and vectors as lines for the dominant component."""
    for C in (33, 64, 65, 97, 128):
        x = np.random.default_rng(C).standard_normal((128, 40, C))
        x[:, :, : C // 2] += np.random.default_rng(1).standard_normal((128, 40, 1))
        m = sc.Multitaper(x, sampling_frequency=128.0, time_halfbandwidth_product=2, n_time_samples_per_window=64)
        vals, vecs = sc.Connectivity.from_multitaper(m).global_coherence(max_rank=3)
        coef, _ = so.multitaper_fft(x, fs=128.0, NW=2, n_time_samples_per_window=64)
        ref, ref_vecs = so.global_coherence(coef, max_rank=3)
        close32(vals, ref, rtol=2e-5, atol_scale=2e-6, what=f"global coherence C={C}")
        np.testing.assert_allclose(np.linalg.norm(vecs, axis=-2), 1.0, atol=1e-9)
        # max_rank = 3 < C - 1: ascending order, the dominant component (shared source) is the last column
        ip = np.abs(np.sum(np.conj(vecs[..., -1]) * ref_vecs[..., -1], axis=-1))
        assert (ip > 1 - 1e-3).mean() > 0.9, f"C={C}: dominant vector differs ({ip.min()})"


@pytest.mark.parametrize("C", [129, 200, 255, 256])
def test_global_coherence_beyond_128_signals(sc, C):
    """129 ... 256 signals: the packed matrix no longer fits LDS and lives in a global scratch (the same Jacobi kernel on
    an L2-resident triangle); values and the dominant vector against the oracle's SVD (float64 engine:
    tests/test_gpu_fp64.py)."""
    x = np.random.default_rng(C).standard_normal((64, 110, C))
    x[:, :, : C // 2] += np.random.default_rng(1).standard_normal((64, 110, 1))
    kw = dict(sampling_frequency=128.0, time_halfbandwidth_product=2, n_time_samples_per_window=32)
    coef, _ = so.multitaper_fft(x, fs=128.0, NW=2, n_time_samples_per_window=32)
    ref, ref_vecs = so.global_coherence(coef, max_rank=3)
    vals, vecs = sc.Connectivity.from_multitaper(sc.Multitaper(x, **kw)).global_coherence(max_rank=3)
    close32(vals, ref, rtol=2e-5, atol_scale=2e-6, what=f"global coherence C={C}")
    np.testing.assert_allclose(np.linalg.norm(vecs, axis=-2), 1.0, atol=1e-9)
    ip = np.abs(np.sum(np.conj(vecs[..., -1]) * ref_vecs[..., -1], axis=-1))
    assert (ip > 1 - 1e-3).mean() > 0.9, f"C={C}: dominant vector differs ({ip.min()})"


@pytest.mark.parametrize("sizes", [(33, 40), (64, 64, 20), (128, 70), (100, 1, 57)])
def test_canonical_coherence_groups_beyond_32_channels(sc, sizes):
    """Groups of 33 ... 128 channels: a workgroup per (bin, group pair) with the blocks in a global scratch and the
    packed Jacobi of sc_jacobi.h, against the oracle's SVD form (n_obs >= every group size)."""
    C = sum(sizes)
    labels = np.repeat(np.arange(len(sizes)), sizes)
    rng = np.random.default_rng(C)
    x = rng.standard_normal((64, 60, C))
    x += 0.7 * rng.standard_normal((64, 60, 1))
    x[:, :, : sizes[0]] += 0.8 * rng.standard_normal((64, 60, 1))
    perm = rng.permutation(C)                                  # groups interleaved over the channel axis
    x, labels = x[:, :, perm], labels[perm]
    kw = dict(sampling_frequency=128.0, time_halfbandwidth_product=2, n_time_samples_per_window=32)
    coef, _ = so.multitaper_fft(x, fs=128.0, NW=2, n_time_samples_per_window=32)
    ref, ref_lab = so.canonical_coherence(coef, labels)
    got, lab = sc.Connectivity.from_multitaper(sc.Multitaper(x, **kw)).canonical_coherence(labels)
    assert np.array_equal(lab, ref_lab)
    close32(got, ref, rtol=5e-5, atol_scale=5e-5, what=f"canonical coherence, groups {sizes}")
    np.testing.assert_array_equal(got, np.swapaxes(got, -1, -2))


def test_wrapper_labelled_outputs_against_the_oracle(sc):
    """multitaper_connectivity() / connectivity_to_xarray() (reference wrapper.py:17-287): dims / coords / names / mt_*
    attributes of the labelled output -- real xarray objects where the package is installed, the vendored minimal
    labelled arrays (spectral_connectivity_amd/_labelled.py) otherwise -- and VALUES against the CPU oracle, not against
    this package's own Connectivity."""
    from spectral_connectivity_amd import multitaper_connectivity
    from spectral_connectivity_amd.wrapper import connectivity_to_xarray
    x = np.random.default_rng(5).standard_normal((400, 6, 3))
    x[:, :, 1] += 0.8 * np.roll(x[:, :, 0], 3, axis=0)
    kw = dict(sampling_frequency=200.0, time_window_duration=0.5, time_halfbandwidth_product=2)
    coef, info = so.multitaper_fft(x, fs=200.0, NW=2, time_window_duration=0.5)
    freqs = so.nonneg_frequencies(np.fft.fftfreq(coef.shape[3], 1 / 200.0))
    da = multitaper_connectivity(x, method="coherence_magnitude", signal_names=["a", "b", "c"], **kw)
    assert da.name == "coherence_magnitude" and tuple(da.dims) == ("time", "frequency", "source", "target")
    assert list(np.asarray(da["source"])) == ["a", "b", "c"] and da.attrs["mt_sampling_frequency"] == 200.0
    assert da.attrs["mt_n_tapers"] == 3 and da.attrs["mt_time_halfbandwidth_product"] == 2
    close32(np.asarray(da.values), so.coherence_magnitude(coef), what="wrapper coherence vs oracle")
    np.testing.assert_allclose(np.asarray(da["frequency"]), freqs)
    np.testing.assert_allclose(np.asarray(da["time"]), np.arange(coef.shape[0]) * 0.5)
    ab = da.sel(source="a", target="b")
    close32(np.asarray(ab.values), so.coherence_magnitude(coef)[..., 0, 1], what="sel(source, target)")
    ds = multitaper_connectivity(x, method=["power", "weighted_phase_lag_index", "imaginary_coherence"], **kw)
    assert set(ds) == {"power", "weighted_phase_lag_index", "imaginary_coherence"}
    assert tuple(ds["power"].dims) == ("time", "frequency", "source")
    close32(np.asarray(ds["power"].values), so.power(coef), what="wrapper power vs oracle")
    close32(np.asarray(ds["weighted_phase_lag_index"].values), so.weighted_phase_lag_index(coef), what="wrapper wpli vs oracle")
    close32(np.asarray(ds["imaginary_coherence"].values), so.imaginary_coherence(coef), what="wrapper imag coh vs oracle")
    two = multitaper_connectivity(x[..., :2], method="coherence_magnitude", squeeze=True, **kw)
    assert tuple(two.dims) == ("time", "frequency")
    coef2, _ = so.multitaper_fft(x[..., :2], fs=200.0, NW=2, time_window_duration=0.5)
    close32(np.asarray(two.values), so.coherence_magnitude(coef2)[..., 0, -1], what="squeezed pair vs oracle")
    one = connectivity_to_xarray(sc.Multitaper(x, **kw), method="phase_locking_value")
    close32(np.asarray(one.values), so.phase_locking_value(coef), rtol=3e-5, atol_scale=3e-5, what="connectivity_to_xarray plv")
    everything = multitaper_connectivity(x, **kw)          # method=None: every expressible measure
    assert {"coherency", "pairwise_spectral_granger_prediction", "phase_locking_value"} <= set(everything)
    assert not ({"group_delay", "global_coherence", "directed_coherence"} & set(everything))


def test_f11_band_statistics_on_the_device_coherency(sc, golden):
    """SURVEY 8(f)-4 on the GPU: phase_slope_index / group_delay / delay of `Connectivity` -- host post-processing of
    the DEVICE coherency -- for the f11 input against the real reference's outputs (tests/golden/f11_post.npz).
    group_delay / delay reproduce the reference by default (NaN / the constants 2 pi k: its one-sample Fisher z is
    always NaN, see options.one_sample_fisher_z)."""
    g = golden("f11_post")
    m = sc.Multitaper(g["x"], sampling_frequency=500.0, time_halfbandwidth_product=3)
    c = sc.Connectivity.from_multitaper(m)
    assert c.n_observations == int(g["n_observations"])
    np.testing.assert_allclose(c.frequencies, g["frequencies"])
    close32(c.coherency(), g["coherency"], what="f11 coherency")
    res = float(g["frequency_resolution"])
    for got, key in ((c.phase_slope_index(), "psi_all"), (c.phase_slope_index([10, 200]), "psi_band"),
                     (c.phase_slope_index([10, 200], res), "psi_band_res")):
        ref = g[key]
        assert got.shape == ref.shape and np.array_equal(np.isnan(got), np.isnan(ref)), key
        ok = ~np.isnan(ref)
        # a sum over ~100 bin pairs of products of f32-accurate coherencies
        assert np.abs(got[ok] - ref[ok]).max() <= 2e-5 * np.abs(ref[ok]).max(), key
    d, s_, r = c.group_delay([10, 200], res)
    for a, key in ((d, "group_delay"), (s_, "group_slope"), (r, "group_r")):
        np.testing.assert_allclose(a, g[key], rtol=1e-12, atol=0, equal_nan=True, err_msg=key)
    np.testing.assert_allclose(c.delay([10, 200], n_range=2), g["delay_band"], rtol=1e-12, atol=0, equal_nan=True)


@pytest.mark.parametrize("C,R", [(2, 9), (16, 7), (32, 30), (40, 5), (64, 4)])
def test_matrix_core_kernel_below_its_crossover(sc, C, R, debug_env):
    """SC_FUSED_NO_SMALL=1 sends every shape through the matrix-core kernel (normally <= 42-58 channels take the f32 VALU
    kernel): the one-block table of <= 32 channels -- a lone MFMA per observation row, whose results the |Im| waves read
    straight away -- and every plane pass (|Im s|, (Im s)^2, sign(Im s)) against the per-plane kernels.  (The sign pass
    once read its MFMA results from inline asm without the wait states the compiler pads for instructions it can see:
    wrong sums at <= 32 channels only.)"""
    from spectral_connectivity_amd import _lib, engine
    debug_env("SC_FUSED_NO_SMALL", "1")
    rng = np.random.default_rng(200 + C)
    x = rng.standard_normal((200, R, C)) + 0.5 * rng.standard_normal((200, R, 1))
    m = sc.Multitaper(x, sampling_frequency=200.0, time_halfbandwidth_product=3,
                      n_time_samples_per_window=64, n_time_samples_per_step=32)
    sp = m.device_spectra()
    for planes, which, tol in ((_lib.PLANE_CSM | _lib.PLANE_ABS_IM, _lib.M_WPLI, 3e-6),
                               (_lib.PLANE_CSM | _lib.PLANE_ABS_IM | _lib.PLANE_IM_SQ, _lib.M_DEBIASED_WPLI2, 2e-5),
                               (_lib.PLANE_SIGN_IM, _lib.M_PLI, None)):
        a_f, n = engine.accumulate(sp, "trials_tapers", planes, use_fused=True)
        a_s, _ = engine.accumulate(sp, "trials_tapers", planes, use_fused=False)
        got = engine.measure(a_f, C, planes, n, which).cpu().numpy()
        ref = engine.measure(a_s, C, planes, n, which).cpu().numpy()
        if tol is None:
            assert np.array_equal(np.isnan(got), np.isnan(ref))
            diff = np.abs(got - ref)[~np.isnan(ref)]
            assert (diff > 0).mean() < 1e-3 and diff.max() <= 8.0 / n, (C, (diff > 0).sum(), diff.max())
        else:
            close32(got, ref, rtol=tol, atol_scale=tol, what=f"C={C} planes {planes:#x}")


@pytest.mark.parametrize("N,L,C,det", [(250, 250, 5, "constant"), (200, 200, 70, "linear"), (300, 250, 33, None),
                                       (1000, 1000, 18, "linear"), (1500, 1500, 6, "constant"), (2000, 1800, 3, "constant"),
                                       (75, 75, 9, "constant"), (120, 100, 2, "linear"), (12, 12, 4, None), (45, 40, 1, "constant"),
                                       (960, 960, 128, "constant"), (500, 500, 17, "linear"), (1280, 1280, 7, None),
                                       (1875, 1875, 4, "constant"), (384, 384, 40, "constant"), (18, 18, 130, "linear")])
def test_mixed_radix_fused_fft_matches_oracle_and_rocfft(sc, N, L, C, det):
    """Window lengths 2^a 3^b 5^c off the power-of-two list (what next_fast_len gives for the usual sampling rates) take
    the mixed-radix fused kernel: against the float64 oracle (zero padding, odd and > 128 channel counts, every detrend),
    and the rocFFT path -- still the transform of every other length -- against the same oracle."""
    import torch
    from spectral_connectivity_amd import _lib, engine
    assert _lib.load().sc_multitaper_fft_supported(L, N) == 1
    rng = np.random.default_rng(N + C)
    R, step = 3, max(L // 2, 1)
    T = L + 2 * step
    x = rng.standard_normal((T, R, C)) + 5.0 + np.linspace(0, 2, T)[:, None, None]
    kw = dict(n_time_samples_per_window=L, n_time_samples_per_step=step, n_fft_samples=N)
    coef, info = so.multitaper_fft(x, fs=200.0, NW=2.5, detrend_type=det, **kw)
    m = sc.Multitaper(x, sampling_frequency=200.0, time_halfbandwidth_product=2.5, detrend_type=det, **kw)
    close32(m.fft(), coef, what=f"mixed-radix N={N}")
    xd = torch.from_numpy(x.astype(np.float32)).cuda()
    h = torch.from_numpy(np.ascontiguousarray(m.tapers.T / 200.0, dtype=np.float32)).cuda()
    sp = engine.multitaper_spectra(xd, h, L, step, N, m.n_time_windows, det, use_fused=False)
    got = np.moveaxis(sp.coefficients().cpu().numpy(), 0, 3)
    close32(got, coef[..., : N // 2 + 1, :], what=f"rocfft N={N}")
    # lengths with another prime factor stay on rocFFT
    assert _lib.load().sc_multitaper_fft_supported(14, 14) == 0 and _lib.load().sc_multitaper_fft_supported(2310, 2310) == 0


@pytest.mark.parametrize("N", [64, 256, 512, 1024, 2048, 200, 250, 1000, 100, 1500, 140])
def test_silent_and_constant_channels_give_exact_zero_spectra(sc, N):
    """The reference transforms every channel on its own, so a silent channel (or a constant one under the default
    constant detrend) has EXACTLY zero coefficients, zero power and exactly zero coherence with every other channel.  The
    fused kernels transform channels in packed pairs: the split must not leave the partner's rounding noise there
    (radix-16, long-window, mixed-radix compile-time / run-time lengths and the rocFFT route, float32 engine)."""
    import torch
    from spectral_connectivity_amd import engine
    rng = np.random.default_rng(N)
    C, R = 6, 4
    x = rng.standard_normal((N, R, C)) * 3.0 + 1.0
    x[:, :, 2] = 0.0
    x[:, :, 5] = 0.75
    m = sc.Multitaper(x, sampling_frequency=100.0, time_halfbandwidth_product=2)
    xd = torch.from_numpy(x.astype(np.float32)).cuda()
    h = torch.from_numpy(np.ascontiguousarray(m.tapers.T / 100.0, dtype=np.float32)).cuda()
    sp = engine.multitaper_spectra(xd, h, N, N, N, 1, "constant")
    X = sp.coefficients().cpu().numpy()                       # (F, W, R, K, C)
    assert np.all(X[..., 2] == 0) and np.all(X[..., 5] == 0), np.abs(X[..., [2, 5]]).max()
    assert np.abs(X[..., [0, 1, 3, 4]]).min() > 0
    c = sc.Connectivity.from_multitaper(m, dtype=np.complex64)
    coh = c.coherence_magnitude()[0]
    off = ~np.eye(C, dtype=bool)
    for ch in (2, 5):                                         # 0 / eps = 0 off the diagonal, NaN on it (connectivity.py:640-670)
        assert np.all(coh[:, ch, off[ch]] == 0) and np.all(coh[:, off[ch], ch] == 0)
    assert np.all(np.isnan(coh[:, np.arange(C), np.arange(C)]))
    ok = [0, 1, 3, 4]
    assert np.all(coh[:, ok][:, :, ok][:, ~np.eye(4, dtype=bool)] > 0)
    assert np.all(c.power()[0][:, [2, 5]] == 0)


def test_large_dc_offset_float32_engine(sc):
    """A DC offset 1e5 times the signal: the float32 engine removes a per-(trial, signal) constant in float64 on the host
    before its float32 cast (every window's detrend removes any constant anyway), so the cast does not eat the signal."""
    rng = np.random.default_rng(5)
    T, R, C = 512, 6, 5
    x = rng.standard_normal((T, R, C))
    x[:, :, 1] += 0.7 * x[:, :, 0]
    x += 1e5 * (1.0 + rng.random((1, R, C)))
    kw = dict(n_time_samples_per_window=128, n_time_samples_per_step=64)
    coef, _ = so.multitaper_fft(x, fs=250.0, NW=3, **kw)
    for det in ("constant", "linear"):
        coef, _ = so.multitaper_fft(x, fs=250.0, NW=3, detrend_type=det, **kw)
        m = sc.Multitaper(x, sampling_frequency=250.0, time_halfbandwidth_product=3, detrend_type=det, **kw)
        c = sc.Connectivity.from_multitaper(m, dtype=np.complex64)
        close32(c.power(), so.power(coef), what=f"power, DC offset, detrend={det}")
        close32(c.coherence_magnitude(), so.coherence_magnitude(coef), rtol=2e-5, atol_scale=2e-5,
                what=f"coherence, DC offset, detrend={det}")


def test_multi_measure_epilogue_equals_single_launches(sc):
    """sc_measure_multi_*: several real-valued measures of one record in one launch, bit-identical to one launch each
    (float32 and float64 output, float and double records, odd channel count, mirrored tiles)."""
    import torch
    from spectral_connectivity_amd import _lib, engine
    rng = np.random.default_rng(17)
    for C, f64 in ((37, False), (64, True)):
        x = rng.standard_normal((256, 9, C))
        kw = dict(sampling_frequency=100.0, time_halfbandwidth_product=2, n_time_samples_per_window=64)
        m = sc.Multitaper(x, **kw)
        sp = m.device_spectra(precision="float64" if f64 else "float32")
        planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM | _lib.PLANE_IM_SQ | _lib.PLANE_SIGN_IM
        accum, n_obs = engine.accumulate(sp, "trials_tapers", planes)
        which = [_lib.M_COHERENCE_MAGNITUDE, _lib.M_WPLI, _lib.M_DEBIASED_WPLI2, _lib.M_PLI]
        for wide in (False, True):
            multi = engine.measure_multi(accum, C, planes, n_obs, which, wide=wide)
            for w, got in zip(which, multi):
                one = engine.measure(accum, C, planes, n_obs, w, wide=wide)
                assert got.dtype == one.dtype and got.shape == one.shape
                assert torch.equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(one, nan=-7.0)), (C, wide, w)
        for grp in ([_lib.M_COHERENCE_PHASE, _lib.M_IMAGINARY_COHERENCE, _lib.M_DEBIASED_PLI2], [_lib.M_PLI, _lib.M_WPLI]):
            for w, got in zip(grp, engine.measure_multi(accum, C, planes, n_obs, grp)):
                one = engine.measure(accum, C, planes, n_obs, w)
                assert torch.equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(one, nan=-7.0)), (C, w)
        # complex measures and power fall back to one launch each
        mixed = engine.measure_multi(accum, C, planes, n_obs, [_lib.M_COHERENCY, _lib.M_WPLI])
        assert mixed[0].is_complex() and not mixed[1].is_complex()


@pytest.mark.parametrize("C", [48, 64, 128, 160])
def test_unit_phasor_plane_with_split_bins(sc, C):
    """PLV / PPC above the small-channel kernel: the staging waves of the matrix-core kernel normalise the rows to unit
    phasors on the way into LDS.  Few bins and many observations, so every bin is split over several workgroups (partial
    records folded by the combine kernel): against a float64 contraction of the same spectra on the device, with a zero
    coefficient (0 / 0 = NaN like the reference) in one channel of one observation."""
    import torch
    from spectral_connectivity_amd import _lib, engine
    F, W, R, K = 5, 2, 700, 3
    g = torch.Generator(device="cuda").manual_seed(C)
    X = torch.view_as_complex(torch.randn((F, W, R, K, C, 2), generator=g, dtype=torch.float32, device="cuda"))
    X[1, 0, 3, 1, 5] = 0
    sp = engine.DeviceSpectra(X, (F, W, R, K, C), (W * R * K * C, R * K * C, K * C, C), 8, True, C_alloc=C)
    planes = _lib.PLANE_UNIT
    accum, n_obs = engine.accumulate(sp, "trials_tapers", planes)
    plv = engine.measure(accum, C, planes, n_obs, _lib.M_PLV).cpu().numpy().reshape(W, F, C, C)
    Xd = X.to(torch.complex128)
    U = Xd / Xd.abs()
    ref = torch.einsum("fwrkc,fwrkd->wfcd", U, U.conj()).abs().cpu().numpy() / n_obs
    bad = np.isnan(ref)
    assert bad[0, 1, 5, :].all() and bad[0, 1, :, 5].all() and bad.sum() == 2 * C - 1
    assert np.array_equal(np.isnan(plv), bad)
    off = ~np.eye(C, dtype=bool)
    np.testing.assert_allclose(plv[:, :, off][~bad[:, :, off]], ref[:, :, off][~bad[:, :, off]], rtol=3e-5, atol=3e-6)


def test_output_dtypes_are_the_references(sc):
    """What the reference returns for every expectation-type measure (observed by running it in the build container,
    numpy 2.2.6): the phase-lag / phase-locking family in the real type of ``dtype``; power and the coherency family in
    the precision of the coefficients (complex128 from Multitaper.fft whatever ``dtype`` is; complex64 coefficients
    handed to the constructor give float32 / complex64)."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal((256, 5, 4))
    m = sc.Multitaper(x, sampling_frequency=100, time_halfbandwidth_product=2)
    family = {"phase_locking_value", "phase_lag_index", "weighted_phase_lag_index", "debiased_squared_phase_lag_index",
              "debiased_squared_weighted_phase_lag_index", "pairwise_phase_consistency"}

    def expect(name, dtype_is_64, coef_is_64):
        wide = dtype_is_64 if name in family else coef_is_64
        if name == "coherency":
            return np.complex128 if wide else np.complex64
        return np.float64 if wide else np.float32

    c64 = sc.Connectivity.from_multitaper(m, dtype=np.complex64)
    c128 = sc.Connectivity.from_multitaper(m)
    raw64 = sc.Connectivity(m.fft().astype(np.complex64), dtype=np.complex64)
    for name in MEASURE_NAMES:
        assert getattr(c64, name)().dtype == expect(name, False, True), name
        assert getattr(c128, name)().dtype == expect(name, True, True), name
        assert getattr(raw64, name)().dtype == expect(name, False, False), name
        close32(getattr(c64, name)(), getattr(c128, name)(), rtol=3e-5, atol_scale=3e-5, what=name)


@pytest.mark.parametrize("L", [64, 256, 250, 1024, 2048, 120, 112])
def test_nonfinite_sample_spoils_only_its_own_channel(sc, L):
    """The fused transforms pack two channels into one complex sequence; a NaN / infinity in one of them must not leak
    into its partner (the reference transforms every channel on its own, transforms.py:1402-1405): the bad channel's
    bins of the windows that contain the sample are NaN, everything else equals the oracle's."""
    rng = np.random.default_rng(L)
    T, R, C = 2 * L, 3, 6
    x = rng.standard_normal((T, R, C))
    x[L // 3, 1, 2] = np.nan            # window 0, trial 1, channel 2 (partner: channel 3)
    x[L + 5, 2, 5] = np.inf             # window 1, trial 2, channel 5 (partner: channel 4)
    with pytest.warns(UserWarning, match="NaN or infinite"):
        m = sc.Multitaper(x, sampling_frequency=500.0, time_halfbandwidth_product=2, n_time_samples_per_window=L)
    got = m.fft()
    clean = np.where(np.isfinite(x), x, 0.0)
    ref, _ = so.multitaper_fft(clean, fs=500.0, NW=2, n_time_samples_per_window=L)
    bad = np.zeros(got.shape, dtype=bool)
    bad[0, 1, :, :, 2] = True
    bad[1, 2, :, :, 5] = True
    assert np.isnan(got[bad]).all()
    assert np.isfinite(got[~bad]).all()
    close32(np.where(bad, 0, got), np.where(bad, 0, ref), what=f"L={L}: channels beside a non-finite one")


def test_f13_canonical_coherence_with_fewer_observations_than_channels(sc, golden):
    """n_trials * n_tapers = 6 against groups of 8 / 6 channels: the reference's SVD form gives 1 for every pair with
    such a group (it spans the whole observation space); the other pairs go through the whitening kernel as usual."""
    g = golden("f13_canonical_few_obs")
    m = sc.Multitaper(g["x"], sampling_frequency=float(g["fs"]), time_halfbandwidth_product=float(g["NW"]))
    c = sc.Connectivity.from_multitaper(m)
    for tag in ("a", "b"):
        cc, _ = c.canonical_coherence(g[f"labels_{tag}"])
        close32(cc, g[f"cc_{tag}"], rtol=2e-5, atol_scale=2e-5, what=f"canonical coherence, few observations ({tag})")


@pytest.mark.parametrize("C,max_rank", [(33, 5), (48, 48), (64, 3), (72, 9), (96, 96), (128, 40), (130, 7), (160, 160)])
def test_global_coherence_any_rank_beyond_64_signals(sc, C, max_rank):
    """max_rank up to n_signals beyond 64 signals (the reference's full SVD, connectivity.py:2245-2279): every value
    against the oracle, and G V = V diag(values) for the returned vectors (each column an eigenvector of the cross-
    spectral matrix with its own value -- a statement that does not depend on how close neighbouring values are)."""
    rng = np.random.default_rng(C + max_rank)
    R = 60
    x = rng.standard_normal((64, R, C)) * (1.0 + 0.3 * np.arange(C))[None, None, :]
    x[:, :, : C // 2] += 2.0 * rng.standard_normal((64, R, 1))
    kw = dict(sampling_frequency=128.0, time_halfbandwidth_product=2, n_time_samples_per_window=32)
    coef, _ = so.multitaper_fft(x, fs=128.0, NW=2, n_time_samples_per_window=32)
    ref, _ = so.global_coherence(coef, max_rank=max_rank)
    vals, vecs = sc.Connectivity.from_multitaper(sc.Multitaper(x, **kw)).global_coherence(max_rank=max_rank)
    assert vals.shape == ref.shape and vecs.shape == ref.shape[:2] + (C, max_rank)
    close32(vals, ref, rtol=3e-5, atol_scale=3e-6, what=f"global coherence C={C} max_rank={max_rank}")
    np.testing.assert_allclose(np.linalg.norm(vecs, axis=-2), 1.0, atol=1e-9)
    W, R_, K, N, _ = coef.shape
    Xm = np.moveaxis(coef.reshape(W, R_ * K, N, C), 1, -1)            # (W, N, C, n_obs)
    G = Xm @ np.conj(np.swapaxes(Xm, -1, -2)) / (R_ * K)
    resid = G @ vecs - vecs * vals[..., None, :]
    scale = np.abs(vals).max()
    assert np.abs(resid).max() <= 3e-5 * scale, f"eigen-residual {np.abs(resid).max():.2e} vs scale {scale:.2e}"


def test_global_coherence_degenerate_eigenvalues_and_the_jacobi_cross_check(sc, debug_env):
    """Beyond 64 signals the eigenpairs come from a Householder tridiagonalisation + bisection + inverse iteration.  Exactly
    repeated eigenvalues (a block-diagonal cross-spectral matrix with two identical blocks) must still give an ORTHONORMAL
    set of eigenvectors (vectors of a cluster are orthogonalised against each other like LAPACK's dstein does), and the
    values must equal those of the parallel-Jacobi kernels of round 2 (SC_GLOBAL_EIG=jacobi keeps them reachable)."""
    rng = np.random.default_rng(5)
    half, n = 40, 60
    z = rng.standard_normal((n, half)) + 1j * rng.standard_normal((n, half))
    z *= (1.0 + np.arange(half))[None, :] ** 0.5
    coef = np.zeros((1, 2 * n, 1, 4, 2 * half), complex)          # (windows, trials, tapers, bins, signals)
    for f in range(4):
        zz = z * np.exp(1j * f)
        coef[0, :n, 0, f, :half] = zz                              # first n observations: block a
        coef[0, n:, 0, f, half:] = zz                              # the others: block b, the same numbers
    K = 12
    c = sc.Connectivity(coef)
    vals, vecs = c.global_coherence(max_rank=K)
    G = np.einsum("oi,oj->ij", coef[0, :, 0, 0, :], coef[0, :, 0, 0, :].conj()) / (2 * n)
    top = np.sort(np.linalg.eigvalsh(G))[::-1][:K][::-1]           # max_rank < C - 1: ascending, like scipy's svds
    np.testing.assert_allclose(vals[0, 0], top, rtol=1e-4 if c._precision == "float32" else 1e-9)
    np.testing.assert_allclose(vals[0, 0, 0::2], vals[0, 0, 1::2], rtol=1e-5)          # every value twice
    V = vecs[0, 0]
    np.testing.assert_allclose(V.conj().T @ V, np.eye(K), atol=1e-6)
    assert np.abs(G @ V - V * vals[0, 0][None, :]).max() <= 1e-4 * np.abs(vals).max()
    debug_env("SC_GLOBAL_EIG", "jacobi")
    vals_j, _ = sc.Connectivity(coef).global_coherence(max_rank=4)
    np.testing.assert_allclose(vals_j, vals[..., -4:], rtol=1e-6)


def test_f14_complex_valued_time_series(sc, golden):
    """Complex series (the reference's generic fft takes them): two-sided coefficients for every detrend mode and the
    measures of the non-negative bins, against vectors from the real reference.  The device transforms the real and the
    imaginary parts as 2 C real series and assembles the two-sided spectrum (Multitaper._complex_device_spectra)."""
    g = golden("f14_complex_series")
    kw = dict(sampling_frequency=float(g["fs"]), time_halfbandwidth_product=float(g["NW"]),
              n_time_samples_per_window=int(g["L"]), n_time_samples_per_step=int(g["step"]))
    for det in ("constant", "linear", None):
        m = sc.Multitaper(g["x"], detrend_type=det, **kw)
        close32(m.fft(), g[f"fft_{det}"], what=f"complex series, detrend={det}")
    c = sc.Connectivity.from_multitaper(m)
    for name in ("power", "coherency", "coherence_magnitude", "weighted_phase_lag_index", "phase_locking_value"):
        close32(getattr(c, name)(), g[name], rtol=3e-5, atol_scale=3e-5, what=f"complex series: {name}")
    granger_close(c.pairwise_spectral_granger_prediction(), g["pairwise_spectral_granger_prediction"], 5e-5,
                  what="complex series: Granger")
