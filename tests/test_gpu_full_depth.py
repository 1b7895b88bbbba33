"""Parity of the FLOAT32 engine (the headline path) at the FULL depth of the BASELINE configurations, elementwise.

The NumPy oracle cannot run configs[2] / configs[4] at full size (its per-observation temporaries are terabytes), so
the oracle's arithmetic is restated in float64 torch on the device (tests/fp64_device_ref.py: torch.fft + einsum, no
product code), PINNED against the NumPy oracle at a reduced trial count inside each test, and then run at the full
size: float64 windows -> detrend -> taper -> FFT -> sum over all observations.  The product path is compared with it
ELEMENTWISE for the outputs north_star names, on BOTH device formats of the float32 engine:

* the planes format (round 4 / 5: stage A writes two f16 pieces per real number, stage B = fused2_kernel: three f16 cross terms
  on the 16-bit matrix pipe + the per-observation |Im s| plane, split-bin partial records summed by the epilogue) -- the route the
  public classes take at these shapes whatever the order of the calls, and the route bench.py times; every such test asserts
  ``c._spectra.P is not None``;
* complex64 spectra (SC_PLANES_FORMAT=0: the kernels of rounds 1-3, fused_csm_absim_kernel = bf16x3 matrix-core CSM with the
  per-observation |Im s| plane; the f32 VALU kernel below ~44 channels), which stay the route of uploaded coefficients.

What float32 arithmetic delivers, measured here and asserted on EVERY entry:  |error| <= 3e-6 |ref| + 2e-7 max|ref|.
The first term is the accumulated f32 rounding of an O(max) value (achieved ~1e-6 on power, 2.5e-6 on wPLI), the second
the noise floor of a cancelling sum of O(max) terms -- coherency of nearly independent channels, the Im S numerator of
wPLI (achieved 3e-8 ... 1.2e-7 of the maximum).  That is 1e-5 RELATIVE on every entry above 3 % of the maximum, and
3e-5 ... 1e-4 relative on entries a thousand times below the maximum (printed).  The 1e-5 relative bar on EVERY entry is met by the float64 engine -- `dtype=complex128`, the reference's
default -- at the same full sizes: tests/test_gpu_fp64.py::test_full_depth_elementwise_relative."""
import numpy as np
import pytest

from oracle import spectral_oracle as so
from fp64_device_ref import measures_fp64, relative_error_report, spectra_fp64, sums_fp64

pytestmark = pytest.mark.gpu
FS = 1000.0
RTOL = 1e-5
FLOOR = 1e-3            # relative errors are REPORTED over the entries above this fraction of the maximum
F32_REL, F32_ABS_OF_MAX = 3e-6, 2e-7       # |err| <= F32_REL |ref| + F32_ABS_OF_MAX max|ref| on every entry


def synth(T, R, C, tone, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((T, R, C)).astype(np.float32)
    t = np.arange(T) / FS
    x += (0.5 * np.sin(2 * np.pi * tone * t[:, None, None] + 2 * np.pi * np.arange(C)[None, None, :] / C)).astype(np.float32)
    return x


@pytest.fixture(scope="module")
def sc():
    import spectral_connectivity_amd as pkg
    return pkg


def pin_against_oracle(x, NW, kw, tapers, L, step, N, names):
    """The float64 torch restatement equals the NumPy oracle (which is pinned to the reference's golden vectors)."""
    coef, _ = so.multitaper_fft(x.astype(np.float64), fs=FS, NW=NW, **kw)
    X = spectra_fp64(x, tapers, FS, L, step, N)
    F = N // 2 + 1
    ref_fft = np.moveaxis(coef[:, :, :, :F, :], 3, 0)                                  # (F, W, R, K, C)
    assert np.abs(X.cpu().numpy() - ref_fft).max() <= 1e-12 * np.abs(ref_fft).max()
    csm, ab = sums_fp64(X, want_abs="weighted_phase_lag_index" in names)
    got = measures_fp64(csm, ab, X.shape[2] * X.shape[3])
    ocsm = so.expectation_csm_gemm(coef)
    ref = dict(power=so.power(coef), coherency=so.coherency(coef, csm=ocsm),
               coherence_magnitude=so.coherence_magnitude(coef, csm=ocsm))
    if "weighted_phase_lag_index" in names:
        ref["weighted_phase_lag_index"] = so.weighted_phase_lag_index(coef)
    for name in names:
        np.testing.assert_allclose(got[name], ref[name], rtol=1e-9, atol=1e-12 * np.nanmax(np.abs(ref[name])),
                                   equal_nan=True, err_msg=name)


def check_elementwise(got, ref, what):
    """|err| <= F32_REL |ref| + F32_ABS_OF_MAX max|ref| on every entry, hence RTOL relative on every entry above 3 % of
    the maximum; the relative error over the entries above FLOOR * max is printed for the record."""
    mx, q999, frac = relative_error_report(got, ref, FLOOR)
    ok = ~np.isnan(ref)
    scale = np.abs(ref[ok]).max()
    err = np.abs(got[ok] - ref[ok])
    worst = (err / (F32_REL * np.abs(ref[ok]) + F32_ABS_OF_MAX * scale)).max()
    cut = F32_ABS_OF_MAX / (RTOL - F32_REL)
    big = np.abs(ref[ok]) > cut * scale
    rel_big = (err[big] / np.abs(ref[ok][big])).max() if big.any() else 0.0
    print(f"  {what}: max abs err / max = {err.max() / scale:.2e}, err / bound = {worst:.2f}; max rel err {rel_big:.2e} "
          f"above {100 * cut:.0f} % of max, {mx:.2e} (99.9th pct {q999:.2e}) over the {100 * frac:.1f} % of entries "
          f"above {FLOOR:g} * max")
    assert worst <= 1.0, f"{what}: err / (3e-6 |ref| + 2e-7 max) = {worst:.2f}"
    assert rel_big <= RTOL
    if what == "power":
        assert mx <= RTOL, f"{what}: elementwise relative error {mx:.3e} > {RTOL:g}"
    return mx


_REF_CACHE = {}


def reference(key, sc, x, NW, kw, names):
    """float64 reference measures of a configuration (computed once per test session: several routes are held to it)."""
    if key not in _REF_CACHE:
        m = sc.Multitaper(x, sampling_frequency=FS, time_halfbandwidth_product=NW, **kw)
        L, step, N = m.n_time_samples_per_window, m.n_time_samples_per_step, m.n_fft_samples
        pin_against_oracle(x[:, :3], NW, kw, m.tapers, L, step, N, names)
        X = spectra_fp64(x, m.tapers, FS, L, step, N)
        csm, ab = sums_fp64(X, want_abs="weighted_phase_lag_index" in names)
        n_obs = X.shape[2] * X.shape[3]
        del X
        _REF_CACHE[key] = (measures_fp64(csm, ab, n_obs), n_obs)
    return _REF_CACHE[key]


def run_config(sc, key, x, NW, kw, names, order=None, planes=None, dtype=None):
    """``order``: the order the measures are requested in (the first request decided the device format before round 5);
    ``planes``: True / False = the route the spectra must have taken (f16 pieces / complex64)."""
    ref, n_obs = reference(key, sc, x, NW, kw, names)
    m = sc.Multitaper(x, sampling_frequency=FS, time_halfbandwidth_product=NW, **kw)
    c = sc.Connectivity.from_multitaper(m) if dtype is None else sc.Connectivity.from_multitaper(m, dtype=dtype)
    assert c.n_observations == n_obs
    got = {name: getattr(c, name)() for name in (order or names)}
    if planes is not None:
        assert (c._spectra.P is not None) == planes, "the spectra took the other device format"
        if planes:
            assert c._spectra._X is None, "the f16 pieces were decoded to complex64: a measure left the planes route"
    return {name: check_elementwise(got[name], ref[name], name) for name in names}, c


CFG2 = (3, {}, ["power", "coherency", "coherence_magnitude", "weighted_phase_lag_index"])
CFG3 = (4, dict(n_time_samples_per_window=256, n_time_samples_per_step=128),
        ["power", "coherency", "coherence_magnitude", "weighted_phase_lag_index"])
CFG5 = (3, {}, ["power", "coherency", "coherence_magnitude"])


def cfg5_series():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1024, 500, 256)).astype(np.float32)
    x += (0.6 * np.repeat(rng.standard_normal((1024, 500, 16)), 16, axis=2)).astype(np.float32)
    return x


def test_cfg2_full_depth_elementwise(sc):
    """configs[1]: 32 ch x 100 trials x 1024 samples, NW = 3, single window, n_obs = 500 (complex64 spectra, the f32 VALU
    stage-B kernel fused_small_kernel: below the 44 channels the planes format starts at)."""
    print("\ncfg2 (32 ch, n_obs 500):")
    run_config(sc, "cfg2", synth(1024, 100, 32, 40.0, 2), *CFG2, planes=False)


def test_cfg3_full_depth_elementwise(sc):
    """configs[2]: 128 ch x 1000 trials x 1024 samples, NW = 4, 256-sample windows step 128: 903 bins x 7000 observations.
    Power first, then coherency, coherence, wPLI: the BASELINE order.  The spectra are f16 pieces; ONE pass of fused2_kernel
    (anticipating the |Im s| plane) serves all four, its three split-bin parts summed by the epilogue."""
    print("\ncfg3 (128 ch, n_obs 7000), planes route, power first:")
    _, c = run_config(sc, "cfg3", synth(1024, 1000, 128, 60.0, 3), *CFG3, planes=True)
    keys = [k for k in c._accum_cache if isinstance(k, int)]
    assert keys == [3], f"one record with CSM + |Im s| was expected, got {keys}"            # PLANE_CSM | PLANE_ABS_IM
    assert c._accum_cache[3][0].dim() == 3, "the split-bin parts were expected unfolded (the path bench.py times)"


def test_cfg3_full_depth_wpli_first_complex64_dtype(sc):
    """The same through Connectivity.from_multitaper(..., dtype=complex64) with wPLI requested FIRST (the order that reached
    fused2_kernel in round 4)."""
    print("\ncfg3 (128 ch, n_obs 7000), planes route, dtype=complex64, wPLI first:")
    order = ["weighted_phase_lag_index", "coherence_magnitude", "coherency", "power"]
    run_config(sc, "cfg3", synth(1024, 1000, 128, 60.0, 3), *CFG3, order=order, planes=True, dtype=np.complex64)


def test_cfg3_full_depth_without_anticipation(sc, monkeypatch):
    """options.anticipate_phase_lag = False: coherence from a CSM-only pass of fused2_kernel, wPLI from a second pass."""
    from spectral_connectivity_amd import options
    monkeypatch.setattr(options, "anticipate_phase_lag", False)
    print("\ncfg3 (128 ch, n_obs 7000), planes route, no anticipation (CSM-only launch, then CSM + |Im s|):")
    _, c = run_config(sc, "cfg3", synth(1024, 1000, 128, 60.0, 3), *CFG3, order=["coherence_magnitude", "weighted_phase_lag_index",
                                                                                 "coherency", "power"], planes=True)
    assert [k for k in c._accum_cache if isinstance(k, int)] == [3]


def test_cfg3_full_depth_complex64_kernels(sc, monkeypatch):
    """SC_PLANES_FORMAT=0: complex64 spectra, fused_csm_absim_kernel (bf16x3 matrix-core CSM + per-observation |Im s|)."""
    monkeypatch.setenv("SC_PLANES_FORMAT", "0")
    print("\ncfg3 (128 ch, n_obs 7000), complex64 route:")
    run_config(sc, "cfg3", synth(1024, 1000, 128, 60.0, 3), *CFG3, planes=False)


def test_cfg3_bench_chain_full_depth(sc):
    """Exactly the chain bench.py's one_step times -- engine.multitaper_spectra(planes_hint = CSM | ABS_IM) ->
    engine.accumulate(fold=False) -> engine.measure_multi([coherence, wPLI]) -- at the full size, against the float64
    reference, and bit for bit against the folded form (fold=True + the same epilogue)."""
    import torch
    from spectral_connectivity_amd import _lib, engine
    x = synth(1024, 1000, 128, 60.0, 3)
    ref, n_obs = reference("cfg3", sc, x, *CFG3)
    m = sc.Multitaper(x, sampling_frequency=FS, time_halfbandwidth_product=4, **CFG3[1])
    h = torch.from_numpy(np.ascontiguousarray(m.tapers.T / FS, dtype=np.float32)).cuda()
    planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
    sp = engine.multitaper_spectra(torch.from_numpy(x).cuda(), h, 256, 128, 256, 7, "constant", planes_hint=planes)
    assert sp.P is not None and sp._X is None
    assert sp.planes_typical_coefficient() > 20.0                       # white noise + a tone: typical coefficients around 2^5 ... 2^6 in scaled units
    accum, n = engine.accumulate(sp, "trials_tapers", planes, fold=False)
    assert n == n_obs == 7000 and accum.dim() == 3 and accum.shape[0] == 3, accum.shape
    coh, wpli = engine.measure_multi(accum, 128, planes, n, [_lib.M_COHERENCE_MAGNITUDE, _lib.M_WPLI])
    folded, _ = engine.accumulate(sp, "trials_tapers", planes)
    coh_f, wpli_f = engine.measure_multi(folded, 128, planes, n, [_lib.M_COHERENCE_MAGNITUDE, _lib.M_WPLI])
    assert torch.equal(engine.fold_parts(accum), folded)
    assert torch.equal(coh.nan_to_num(), coh_f.nan_to_num()) and torch.equal(wpli, wpli_f)
    print("\ncfg3 bench chain (planes stage A -> fused2 parts -> parts-summing epilogue):")
    check_elementwise(coh.reshape(7, 129, 128, 128).cpu().numpy().astype(np.float64), ref["coherence_magnitude"], "coherence_magnitude")
    check_elementwise(wpli.reshape(7, 129, 128, 128).cpu().numpy().astype(np.float64), ref["weighted_phase_lag_index"], "weighted_phase_lag_index")


def test_cfg5_full_depth_elementwise(sc):
    """configs[4] shape: 256 ch x 500 trials x 1024 samples, NW = 3, single window: 513 bins x 2500 observations through the
    planes route (fused2_kernel's six staircase launches over 32-channel blocks; 1024-sample planes transform)."""
    print("\ncfg5 (256 ch, n_obs 2500), planes route:")
    run_config(sc, "cfg5", cfg5_series(), *CFG5, planes=True)


def test_cfg5_full_depth_complex64_kernels(sc, monkeypatch):
    """The same on complex64 spectra (the kernel behind canonical coherence's records at this shape)."""
    monkeypatch.setenv("SC_PLANES_FORMAT", "0")
    print("\ncfg5 (256 ch, n_obs 2500), complex64 route:")
    run_config(sc, "cfg5", cfg5_series(), *CFG5, planes=False)


@pytest.mark.parametrize("route", ["planes", "complex64"])
def test_cfg3_abs_im_plane_full_depth(sc, route):
    """The sum |Im s| record plane itself (the wPLI weights) at configs[2] full size against float64 sums of the SAME
    device spectra: isolates stage B (per-observation products on the matrix cores, f32 |d| accumulation over 7000 observations,
    split bins) from the f32 transform -- fused2_kernel on the f16 pieces (compared with the float64 sums of their decoded
    values) and fused_csm_absim_kernel on complex64."""
    import torch
    from spectral_connectivity_amd import _lib, engine
    x = synth(1024, 1000, 128, 60.0, 3)
    kw = dict(n_time_samples_per_window=256, n_time_samples_per_step=128)
    m = sc.Multitaper(x, sampling_frequency=FS, time_halfbandwidth_product=4, **kw)
    planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
    sp = m.device_spectra(planes_hint=planes if route == "planes" else None)
    assert (sp.P is not None) == (route == "planes")
    X = sp.X.reshape(129, 7, 1000, 7, 128).to(torch.complex128)          # (planes: decoded from the pieces, exactly)
    csm, ab = sums_fp64(X)
    del X
    if route == "planes":
        sp._X = None                                                     # stage B must read the pieces
    accum, n = engine.accumulate(sp, "trials_tapers", planes)
    assert n == 7000
    ref = measures_fp64(csm, ab, n)
    got_w = engine.measure(accum, 128, planes, n, _lib.M_WPLI).reshape(7, 129, 128, 128).cpu().numpy().astype(np.float64)
    got_s = engine.measure(accum, 128, planes, n, _lib.M_CSM).reshape(7, 129, 128, 128).cpu().numpy()
    print(f"\ncfg3 stage B alone, {route} (same spectra on both sides):")
    check_elementwise(got_w, ref["weighted_phase_lag_index"], "wPLI")
    # the weight plane itself, decoded from the record: A[bin][plane][tile][16][16], upper-triangular 16 x 16 tiles
    # (tile index = bi * NB - bi (bi - 1) / 2 + (bj - bi)), plane 2 = sum |Im s| when planes = CSM | ABS_IM
    NB = 8
    rec = accum.reshape(7 * 129, 3, NB * (NB + 1) // 2, 16, 16)[:, 2].cpu().numpy().astype(np.float64)
    w_got = np.zeros((7 * 129, 128, 128))
    for bi in range(NB):
        for bj in range(bi, NB):
            w_got[:, 16 * bi:16 * bi + 16, 16 * bj:16 * bj + 16] = rec[:, bi * NB - bi * (bi - 1) // 2 + (bj - bi)]
    w_got = w_got.reshape(7, 129, 128, 128) / n
    w_ref = (ab / n).cpu().numpy()
    upper = np.triu(np.ones((128, 128), dtype=bool), 1)
    inner = slice(1, 128)                                              # DC / Nyquist: Im s = 0 exactly
    rel = np.abs(w_got[:, inner][..., upper] - w_ref[:, inner][..., upper]) / w_ref[:, inner][..., upper]
    print(f"  sum |Im s| plane (positive terms): max rel err {rel.max():.2e}, 99.9th pct {np.quantile(rel, 0.999):.2e}")
    assert rel.max() <= 1e-5
    im_ref = (csm.imag / n).cpu().numpy()
    err_im = np.abs(got_s.imag - im_ref).max() / np.abs(im_ref).max()
    assert err_im < 3e-6, err_im
