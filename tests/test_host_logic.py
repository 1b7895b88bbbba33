"""Host-side mirror of the reference API: derived parameters, tapers, validation messages.
Expected values are golden vectors produced by the real reference (oracle/gen_golden.py);
the error/warning regexes are the ones the reference's own tests match
(reference tests/test_transforms.py:338-586, tests/test_connectivity.py:855-895)."""
import warnings

import numpy as np
import pytest

from spectral_connectivity_amd import Connectivity, Multitaper, estimate_n_tapers, prepare_time_series
from spectral_connectivity_amd.transforms import dpss_windows


def test_geometry_matches_reference(golden):
    g = golden("f4_lengths")
    x = g["x"]
    cases = {
        "L250": dict(n_time_samples_per_window=250),
        "L250_N300": dict(n_time_samples_per_window=250, n_fft_samples=300),
        "L255": dict(n_time_samples_per_window=255),
        "L256_N255": dict(n_time_samples_per_window=256, n_fft_samples=255),
        "dur_step": dict(time_window_duration=0.8, time_window_step=0.29),
    }
    for tag, kw in cases.items():
        m = Multitaper(x, sampling_frequency=250.0, time_halfbandwidth_product=2, **kw)
        np.testing.assert_allclose(m.time, g[f"{tag}__time"], rtol=1e-12)
        np.testing.assert_allclose(m.frequencies, g[f"{tag}__frequencies"], rtol=1e-12)
        assert m.n_fft_samples == g[f"{tag}__fft"].shape[3]
        assert m.n_time_windows == g[f"{tag}__fft"].shape[0]
        c = Connectivity(np.zeros((1, 1, 1, m.n_fft_samples, 2), complex), frequencies=m.frequencies)
        np.testing.assert_allclose(c.frequencies, g[f"{tag}__conn_frequencies"], rtol=1e-12)


def test_step_truncation_quirk():
    # int(0.29*100) == 28: the reference truncates the step but rounds the window
    m = Multitaper(np.zeros((1000, 1, 1)), sampling_frequency=100, time_window_duration=0.5,
                   time_window_step=0.29)
    assert m.n_time_samples_per_step == 28 and m.n_time_samples_per_window == 50


@pytest.mark.parametrize("L,NW", [(1024, 3.0), (256, 4.0), (4096, 3.0), (250, 2.0), (64, 2.5)])
def test_dpss_matches_reference(golden, L, NW):
    g = golden("f8_dpss")
    tapers, eig = dpss_windows(L, NW, int(np.floor(2 * NW - 1)), is_low_bias=False)
    np.testing.assert_allclose(tapers, g[f"L{L}_NW{NW}__tapers"], rtol=1e-7, atol=1e-11)
    np.testing.assert_allclose(eig, g[f"L{L}_NW{NW}__eig"], rtol=1e-9)


def test_tapers_property_and_low_bias(golden):
    g = golden("f7_edges")
    x = g["nw175__x"]
    m = Multitaper(x, sampling_frequency=100.0, time_halfbandwidth_product=1.75)
    np.testing.assert_allclose(m.tapers, g["nw175__tapers"], atol=1e-10)
    m = Multitaper(x, sampling_frequency=100.0, time_halfbandwidth_product=1.0)
    np.testing.assert_allclose(m.tapers, g["nw1__tapers"], atol=1e-10)
    # reference tests/test_transforms.py:62-71
    assert [estimate_n_tapers(nw) for nw in (3, 1, 1.75)] == [5, 1, 2]
    assert Multitaper(x, time_halfbandwidth_product=3).n_tapers == 5


def test_multitaper_validation_messages():
    with pytest.raises(ValueError, match=r"Expected 3D array.*got 1D"):
        Multitaper(np.zeros(10))
    with pytest.raises(ValueError, match=r"Expected 3D array.*got 2D"):
        Multitaper(np.zeros((10, 2)))
    with pytest.raises(ValueError, match=r"Expected 3D array.*got 4D"):
        Multitaper(np.zeros((10, 2, 2, 2)))
    x = np.zeros((100, 2, 2))
    with pytest.raises(ValueError, match=r"sampling_frequency.*must be positive"):
        Multitaper(x, sampling_frequency=0)
    with pytest.raises(ValueError, match=r"time_halfbandwidth_product.*must be at least 1"):
        Multitaper(x, time_halfbandwidth_product=0.5)
    with pytest.raises(ValueError, match=r"time_window_duration.*must be positive"):
        Multitaper(x, time_window_duration=-1)
    with pytest.raises(ValueError, match=r"time_window_step.*must be positive"):
        Multitaper(x, time_window_step=0)
    with pytest.warns(UserWarning, match=r"data may be transposed"):
        Multitaper(np.zeros((5, 1, 10)))
    bad = x.copy()
    bad[3, 0, 0] = np.nan
    with pytest.warns(UserWarning, match=r"contains NaN.*infinite values"):
        Multitaper(bad)
    with pytest.warns(UserWarning, match=r"unusually large"):
        Multitaper(x, time_halfbandwidth_product=11)
    with pytest.warns(UserWarning, match=r"creates gaps"):
        Multitaper(x, sampling_frequency=100, time_window_duration=0.1, time_window_step=0.2)


def test_prepare_time_series():
    assert prepare_time_series(np.zeros(7)).shape == (7, 1, 1)
    assert prepare_time_series(np.zeros((7, 3)), axis="signals").shape == (7, 1, 3)
    assert prepare_time_series(np.zeros((7, 3)), axis="trials").shape == (7, 3, 1)
    with pytest.raises(ValueError, match=r"For 2D input.*must specify.*axis.*parameter"):
        prepare_time_series(np.zeros((7, 3)))
    with pytest.raises(ValueError, match=r"axis must be.*'signals'.*'trials'"):
        prepare_time_series(np.zeros((7, 3)), axis="x")


def test_connectivity_validation_messages():
    for nd in (1, 2, 3, 4, 6):
        with pytest.raises(ValueError, match=f"must be 5-dimensional, got {nd}D"):
            Connectivity(np.zeros((2,) * nd, complex))
    with pytest.raises(ValueError, match=r"Expected shape.*n_time_windows.*n_trials.*n_tapers"):
        Connectivity(np.zeros((2, 2), complex))
    with pytest.raises(ValueError, match="use the Multitaper class"):
        Connectivity(np.zeros((2, 2), complex))
    with pytest.raises(ValueError, match=r"Did you mean 'trials_tapers'"):
        Connectivity(np.zeros((1, 1, 1, 4, 2), complex), expectation_type="tapers_trials")
    coef = np.zeros((1, 1, 1, 4, 2), complex)
    coef[0, 0, 0, 0, 0] = np.inf
    with pytest.warns(UserWarning, match="NaN or Inf"):
        Connectivity(coef)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        c = Connectivity(np.ones((3, 5, 7, 8, 2), complex), expectation_type="time_tapers")
    assert c.n_observations == 21


def test_parameter_helpers_match_reference_values():
    """Expected dicts were produced by the real reference (reference transforms.py:199-402)."""
    from spectral_connectivity_amd import suggest_parameters
    assert suggest_parameters(1000, 10.0) == {
        "sampling_frequency": 1000, "time_halfbandwidth_product": 3.0, "time_window_duration": 2.0,
        "n_tapers": 5, "frequency_resolution": 3.0, "n_time_windows": 5, "nyquist_frequency": 500.0}
    p = suggest_parameters(500, 60.0, desired_n_tapers=9)
    assert (p["time_halfbandwidth_product"], p["n_tapers"], p["time_window_duration"]) == (5.0, 9, 12.0)
    p = suggest_parameters(1000, 3.0, desired_freq_resolution=3.0)      # window capped at a third of the signal
    assert p["time_window_duration"] == 1.0 and p["time_halfbandwidth_product"] == 1.5 and p["n_time_windows"] == 3
    with pytest.raises(ValueError, match="Cannot achieve desired frequency resolution"):
        suggest_parameters(250, 2.0, desired_freq_resolution=0.5)
    with pytest.warns(UserWarning, match="Both 'desired_freq_resolution' and 'desired_n_tapers'"):
        suggest_parameters(1000, 10.0, desired_freq_resolution=2.0, desired_n_tapers=3)
    m = Multitaper(np.zeros((5000, 1, 64)), sampling_frequency=1000, time_window_duration=1.0,
                   time_halfbandwidth_product=3)
    text = m.summarize_parameters()
    for line in ("Time samples:    5000 (5.00 seconds)", "Number of tapers:              5",
                 "Window step:      1.000 s (non-overlapping)", "Number of windows: 5",
                 "Frequency resolution: 6.0 Hz", "FFT samples:          1000"):
        assert line in text, line


# ---- band post-processing of the coherency and the statistics helpers (host-side NumPy) ------------------
def test_phase_slope_index_and_delay_match_reference(golden):
    from spectral_connectivity_amd import _postprocess as pp
    g = golden("f11_post")
    coh, f, res = g["coherency"], g["frequencies"], float(g["frequency_resolution"])
    np.testing.assert_allclose(pp.phase_slope_index(coh, f), g["psi_all"], rtol=1e-9, atol=1e-9, equal_nan=True)
    np.testing.assert_allclose(pp.phase_slope_index(coh, f, [10, 200]), g["psi_band"], rtol=1e-9, atol=1e-9, equal_nan=True)
    np.testing.assert_allclose(pp.phase_slope_index(coh, f, [10, 200], res), g["psi_band_res"], rtol=1e-9, atol=1e-9,
                               equal_nan=True)
    # delay(): the reference's output is the constant 2 pi k for every frequency and pair (raw data of a fully
    # masked array, because its one-sample z-score is always NaN) -- reproduced by default
    ref = g["delay_band"]
    np.testing.assert_allclose(ref[0, :, :, 0, 1], np.broadcast_to(2 * np.pi * np.arange(-2, 3), ref.shape[1:3]))
    got = pp.delay(coh, f, int(g["n_observations"]), [10, 200], n_range=2)
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=0, equal_nan=True)
    d, s_, r = pp.group_delay(coh, f, int(g["n_observations"]), [10, 200], res)
    for a, b in ((d, g["group_delay"]), (s_, g["group_slope"]), (r, g["group_r"])):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=0, equal_nan=True)


@pytest.fixture
def unbiased_one_sample_z():
    from spectral_connectivity_amd import options
    options.one_sample_fisher_z = "unbiased"
    yield
    options.one_sample_fisher_z = "reference"


def test_delay_with_the_unbiased_one_sample_z(golden, unbiased_one_sample_z):
    """options.one_sample_fisher_z = "unbiased": delay() carries the candidates (phase + 2 pi k) / 2 pi at the
    significant frequencies and NaN elsewhere."""
    from spectral_connectivity_amd import _postprocess as pp
    g = golden("f11_post")
    coh, f = g["coherency"], g["frequencies"]
    got = pp.delay(coh, f, int(g["n_observations"]), [10, 200], n_range=2)
    assert got.shape == g["delay_band"].shape
    band = f[(f > 10) & (f < 200)]
    k0 = got[0, :, 2, 0, 1]                      # k = 0 candidate, pair (0, 1): phase / 2 pi = f * tau
    ok = ~np.isnan(k0)
    assert ok.sum() >= 20
    assert abs(np.median(k0[ok] / band[ok]) - 0.010) < 3e-4 and np.abs(k0[ok] / band[ok] - 0.010).max() < 6e-3
    np.testing.assert_allclose(got[0, :, 3, 0, 1][ok] - k0[ok], 1.0)
    np.testing.assert_allclose(got[0, :, :, 1, 0], -got[0, :, :, 0, 1], equal_nan=True)


def test_group_delay_recovers_a_known_delay(golden, unbiased_one_sample_z):
    """Channel 1 is channel 0 delayed by 5 samples at 500 Hz.  The reference's own group_delay() returns NaN
    for every pair (its one-sample Fisher z evaluates coherence_bias(0) = -1/2 and takes the square root of a
    negative number, so no frequency is ever significant) -- the default here as well; with
    options.one_sample_fisher_z = "unbiased" the regression recovers the delay."""
    from spectral_connectivity_amd import _postprocess as pp
    g = golden("f11_post")
    assert np.isnan(g["group_delay"][0, 0, 1]) and np.isnan(g["group_r"][0, 0, 1]) and g["group_r"][0, 0, 0] == 1.0
    d, s, r = pp.group_delay(g["coherency"], g["frequencies"], int(g["n_observations"]), [10, 200],
                             float(g["frequency_resolution"]))
    assert d.shape == g["group_delay"].shape
    assert abs(d[0, 0, 1] - 0.010) < 2e-4 and abs(d[0, 1, 0] + 0.010) < 2e-4
    assert r[0, 0, 1] > 0.999 and r[0, 0, 0] == 1.0 and np.isnan(s[0, 0, 0])
    np.testing.assert_allclose(s, 2 * np.pi * d, equal_nan=True)
    # a pair of independent channels has no significant run: NaN
    assert np.isnan(d[0, 0, 2]) or abs(r[0, 0, 2]) <= 1.0


def test_statistics_helpers_match_reference(golden):
    from spectral_connectivity_amd import statistics as st
    g = golden("f11_post")
    p = g["stat_p"]
    np.testing.assert_array_equal(st.Benjamini_Hochberg_procedure(p, alpha=0.05), g["stat_bh"])
    np.testing.assert_array_equal(st.Benjamini_Hochberg_procedure(0.5 + 0.5 * p, alpha=0.01), g["stat_bh_none"])
    np.testing.assert_array_equal(st.Bonferroni_correction(p, alpha=0.05), g["stat_bonf"])
    np.testing.assert_array_equal(st.adjust_for_multiple_comparisons(p, method="Bonferroni_correction"), g["stat_bonf"])
    z = st.coherence_fisher_z_transform(g["stat_coh1"], 40, g["stat_coh2"], 25)
    np.testing.assert_allclose(z, g["stat_fisher2"], rtol=1e-12)
    np.testing.assert_allclose(st.get_normal_distribution_p_values(z), g["stat_pvals"], rtol=1e-12)
    assert st.coherence_bias(40) == float(g["stat_coh_bias"])
    np.testing.assert_allclose(st.coherence_rate_adjustment(10.0, 14.0, np.linspace(0.5, 3, 6), 0.2, 0.5),
                               g["stat_rate_adj"], rtol=1e-12)
    lo, hi = st.power_confidence_intervals(7, power=np.linspace(1, 4, 5), ci=0.9)
    np.testing.assert_allclose(lo, g["stat_ci_lo"], rtol=1e-12)
    np.testing.assert_allclose(hi, g["stat_ci_hi"], rtol=1e-12)
    np.testing.assert_allclose(st.power_bias(35), g["stat_power_bias"], rtol=1e-12)
    np.testing.assert_allclose(st.power_variance(35), g["stat_power_var"], rtol=1e-12)
    np.testing.assert_allclose(st.power_fisher_z_transform(np.linspace(1, 4, 5), 35, np.linspace(2, 3, 5), 21),
                               g["stat_power_z"], rtol=1e-12)
    # one-sample z-score: NaN like the reference's by default, finite with the unbiased option
    from spectral_connectivity_amd import options
    assert np.isnan(st.coherence_fisher_z_transform(g["stat_coh1"], 40)).all()
    options.one_sample_fisher_z = "unbiased"
    try:
        assert np.isfinite(st.coherence_fisher_z_transform(g["stat_coh1"], 40)).all()
    finally:
        options.one_sample_fisher_z = "reference"


def test_wrapper_validates_before_touching_the_device():
    """Methods the labelled interface cannot express are refused with the reference's message (wrapper.py:71-78)
    -- before xarray or the GPU are needed."""
    from spectral_connectivity_amd import multitaper_connectivity
    from spectral_connectivity_amd.wrapper import connectivity_to_xarray
    x = np.random.default_rng(0).standard_normal((64, 2, 2))
    for bad in ("group_delay", "canonical_coherence", "directed_transfer_function", "partial_directed_coherence"):
        with pytest.raises(ValueError, match="not supported by the xarray interface"):
            multitaper_connectivity(x, sampling_frequency=100, method=bad)
        with pytest.raises(ValueError, match="Connectivity class directly"):
            connectivity_to_xarray(Multitaper(x, sampling_frequency=100), method=bad)


def test_vendored_labelled_arrays():
    """_labelled.DataArray / Dataset: what the front end returns when the optional xarray package is absent."""
    from spectral_connectivity_amd._labelled import DataArray, Dataset
    v = np.arange(2 * 3 * 2 * 2, dtype=float).reshape(2, 3, 2, 2)
    a = DataArray(v, coords=[[0.0, 0.5], [0.0, 10.0, 20.0], ["a", "b"], ["a", "b"]],
                  dims=["time", "frequency", "source", "target"], name="coherence_magnitude", attrs={"mt_n_tapers": 5})
    assert a.dims == ("time", "frequency", "source", "target") and a.shape == (2, 3, 2, 2) and a.name == "coherence_magnitude"
    np.testing.assert_array_equal(a["frequency"], [0.0, 10.0, 20.0])
    np.testing.assert_array_equal(a.sel(source="a", target="b").values, v[:, :, 0, 1])
    assert a.sel(source="a", target="b").dims == ("time", "frequency")
    np.testing.assert_array_equal(a.sel(frequency=12.0, method="nearest").values, v[:, 1])
    np.testing.assert_array_equal(a.isel(time=1, frequency=[0, 2]).values, v[1][[0, 2]])
    assert a.isel(time=[0]).squeeze().dims == ("frequency", "source", "target")
    np.testing.assert_array_equal(np.asarray(a), v)
    with pytest.raises(KeyError):
        a.sel(source="zz")
    with pytest.raises(ValueError, match="coordinate 'frequency'"):
        DataArray(v, coords=[[0, 1], [0, 1], ["a", "b"], ["a", "b"]], dims=["time", "frequency", "source", "target"])
    ds = Dataset()
    ds["coherence_magnitude"] = a
    assert list(ds) == ["coherence_magnitude"] and ds.data_vars is ds and ds.attrs == {"mt_n_tapers": 5}


def test_host_detrend_helper_matches_scipy_and_reference_messages():
    """transforms.detrend (reference transforms.py:1798-1915, a restatement of scipy.signal.detrend)."""
    import scipy.signal
    from spectral_connectivity_amd.transforms import detrend
    x = np.random.default_rng(0).standard_normal((50, 3, 7)) + np.linspace(0, 5, 50)[:, None, None]
    for kw in (dict(axis=0), dict(axis=0, bp=[20, 35]), dict(axis=0, type="constant"), dict(axis=-1), dict(axis=1, type="l")):
        np.testing.assert_allclose(detrend(x, **kw), scipy.signal.detrend(x, **kw), atol=1e-12)
    with pytest.raises(ValueError, match="Invalid trend type 'cubic' is not supported"):
        detrend(x, type="cubic")
    with pytest.raises(ValueError, match="exceed data length"):
        detrend(np.zeros(100), type="linear", bp=[150])


def test_simulate_mvar_reproduces_a_known_var_process():
    """simulate.simulate_MVAR: shape contract, determinism in the seed, and the lag-1 cross-covariance of a stable
    VAR(1) (Gamma_1 = A Gamma_0) on a long realisation."""
    from spectral_connectivity_amd.simulate import simulate_MVAR
    A = np.array([[[0.5, 0.2], [-0.1, 0.4]]])
    x = simulate_MVAR(A, n_time_samples=20000, n_trials=2, n_burnin_samples=200, random_state=1)
    assert x.shape == (20000, 2, 2)
    np.testing.assert_array_equal(x, simulate_MVAR(A, n_time_samples=20000, n_trials=2, n_burnin_samples=200,
                                                   random_state=np.random.default_rng(1)))
    z = x[:, 0]
    g0 = z[:-1].T @ z[:-1] / (len(z) - 1)
    g1 = z[1:].T @ z[:-1] / (len(z) - 1)
    np.testing.assert_allclose(g1, A[0] @ g0, atol=0.05)


def test_dpss_interpolated_from_a_shorter_window():
    """dpss_windows(interp_from=...): tapers of the short length, interpolated and renormalised (transforms.py:1615-1651);
    they stay close to the directly computed ones, keep unit norm and the sign conventions."""
    direct, eig = dpss_windows(256, 3, 5, is_low_bias=False)
    interp, eig_i = dpss_windows(256, 3, 5, is_low_bias=False, interp_from=64, interp_kind="cubic")
    assert interp.shape == direct.shape
    np.testing.assert_allclose(np.linalg.norm(interp, axis=1), 1.0, atol=1e-12)
    assert np.abs(interp - direct).max() < 2e-2 and np.abs(eig - eig_i).max() < 2e-2     # the grid is shifted (endpoint=False)
    assert (interp[::2].sum(axis=1) > 0).all()


def test_tridiagonal_helpers():
    """tridisolve / tridi_inverse_iteration (reference transforms.py:1443-1536): solve and eigenvector of a symmetric
    tridiagonal matrix, checked against dense linear algebra."""
    from spectral_connectivity_amd.transforms import tridi_inverse_iteration, tridisolve
    rng = np.random.default_rng(0)
    d, e, b = rng.uniform(2, 3, 9), rng.uniform(-0.5, 0.5, 8), rng.standard_normal(9)
    A = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    np.testing.assert_allclose(tridisolve(d, e, b.copy()), np.linalg.solve(A, b), rtol=1e-12)
    keep = b.copy()
    out = tridisolve(d, e, keep, overwrite_b=False)
    np.testing.assert_array_equal(keep, b)
    np.testing.assert_allclose(out, np.linalg.solve(A, b), rtol=1e-12)
    w, V = np.linalg.eigh(A)
    v = tridi_inverse_iteration(d, e, w[-1] + 1e-9, x0=np.ones(9))
    assert abs(abs(v @ V[:, -1]) - 1.0) < 1e-8 and abs(np.linalg.norm(v) - 1.0) < 1e-12


def test_public_api_surface_matches_the_reference():
    """Every public function, class, method / property and module constant of the reference exists here with the same
    argument names, order and default values (tests/golden/api_surface.json, generated from the reference by
    oracle/gen_golden.py api).  Only private helpers differ."""
    import ast
    import json
    import os
    import spectral_connectivity_amd as pkg
    root = os.path.dirname(pkg.__file__)
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "api_surface.json")))
    assert set(ref["__all__"]) <= set(pkg.__all__)

    def describe(fn):
        a = fn.args
        pos = a.posonlyargs + a.args
        dflt = [None] * (len(pos) - len(a.defaults)) + [ast.unparse(x) for x in a.defaults]
        args = [[p.arg, d] for p, d in zip(pos, dflt)]
        args += [[k.arg, ast.unparse(v) if v is not None else None] for k, v in zip(a.kwonlyargs, a.kw_defaults)]
        return args

    problems = []
    for module, names in ref.items():
        if module == "__all__":
            continue
        path = os.path.join(root, module + ".py")
        assert os.path.exists(path), f"module {module} missing"
        tree = ast.parse(open(path).read())
        have = {}
        for n in tree.body:
            if isinstance(n, ast.FunctionDef):
                have[n.name] = describe(n)
            elif isinstance(n, ast.ClassDef):
                for m in n.body:
                    if isinstance(m, ast.FunctionDef):
                        have[n.name + "." + m.name] = describe(m)
            elif isinstance(n, ast.Assign):
                for tg in n.targets:
                    if isinstance(tg, ast.Name):
                        have[tg.id] = ast.unparse(n.value)
        for name, spec in names.items():
            if name not in have:
                problems.append(f"{module}.{name} missing")
            elif "constant" in spec:
                if name == "TIKHONOV_REGULARIZATION_FACTOR":
                    assert float(have[name]) == float(spec["constant"])
            else:
                mine = [[a, (d or "").replace("np.", "xp.") or None] for a, d in have[name]]
                want = [[a, (d or "").replace("np.", "xp.") or None] for a, d in spec["args"]]
                if mine != want:
                    problems.append(f"{module}.{name}: {mine} != {want}")
    assert not problems, "\\n".join(problems)
