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
