"""Range / symmetry properties of every measure on the device path (the reference pins the same properties in
tests/test_metric_ranges.py:20-153 and tests/test_coherence_bounds.py:6-78 with this fixture: 100 samples x 5 trials
x 3 signals of sine + noise, seed 42, NW = 2, 3 tapers)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def conn():
    import spectral_connectivity_amd as sc
    rng = np.random.default_rng(42)
    t = np.arange(100) / 100.0
    x = 0.5 * rng.standard_normal((100, 5, 3))
    x += np.sin(2 * np.pi * 10 * t)[:, None, None] * np.array([1.0, 0.7, 0.4])[None, None, :]
    m = sc.Multitaper(x, sampling_frequency=100.0, time_halfbandwidth_product=2, n_tapers=3)
    return sc.Connectivity.from_multitaper(m)


def _offdiag(a):
    c = a.shape[-1]
    return a[..., ~np.eye(c, dtype=bool)]


@pytest.mark.parametrize("name", ["coherence_magnitude", "imaginary_coherence", "phase_locking_value",
                                  "directed_transfer_function", "partial_directed_coherence",
                                  "generalized_partial_directed_coherence", "direct_directed_transfer_function"])
def test_unit_interval_measures(conn, name):
    v = _offdiag(getattr(conn, name)())
    assert np.isfinite(v).all() and (v >= 0).all() and (v <= 1 + 1e-6).all()


@pytest.mark.parametrize("name", ["phase_lag_index", "weighted_phase_lag_index"])
def test_signed_unit_interval_and_antisymmetry(conn, name):
    v = getattr(conn, name)()
    assert np.isfinite(_offdiag(v)).all() and (np.abs(_offdiag(v)) <= 1 + 1e-6).all()
    np.testing.assert_allclose(v, -np.swapaxes(v, -1, -2), atol=1e-6)


def test_power_positive_and_coherency_hermitian(conn):
    assert (conn.power() > 0).all()
    c = conn.coherency()
    assert np.isnan(np.diagonal(c, axis1=-1, axis2=-2)).all()
    o = np.where(np.isnan(c), 0, c)
    np.testing.assert_allclose(o, np.conj(np.swapaxes(o, -1, -2)), atol=1e-6)
    np.testing.assert_allclose(np.abs(_offdiag(c)) ** 2, _offdiag(conn.coherence_magnitude()), rtol=2e-5, atol=1e-6)
    ph = conn.coherence_phase()
    assert (np.abs(_offdiag(ph)) <= np.pi + 1e-6).all()


def test_debiased_and_consistency_measures_are_finite(conn):
    for name in ("debiased_squared_phase_lag_index", "debiased_squared_weighted_phase_lag_index",
                 "pairwise_phase_consistency"):
        # interior bins: at DC and Nyquist every Im s of a real signal is exactly 0, and the debiased wPLI is 0/0 = NaN
        # there by the reference's own formula (connectivity.py:1060-1127)
        v = _offdiag(getattr(conn, name)()[:, 1:-1])
        assert np.isfinite(v).all() and (v <= 1 + 1e-6).all()


def test_normalisations_of_the_mvar_measures(conn):
    dtf = conn.directed_transfer_function()
    np.testing.assert_allclose(dtf.sum(axis=-1), 1.0, atol=1e-9)          # inflow into every node sums to one
    pdc = conn.partial_directed_coherence()
    np.testing.assert_allclose(pdc.sum(axis=-2), 1.0, atol=1e-9)          # outflow of every node sums to one
    gpdc = conn.generalized_partial_directed_coherence()
    np.testing.assert_allclose(gpdc.sum(axis=-2), 1.0, atol=1e-9)
    sigma = conn._noise_covariance
    np.testing.assert_allclose(sigma, np.swapaxes(sigma, -1, -2), atol=1e-12)
    assert (np.linalg.eigvalsh(sigma) > 0).all()
    # H A = I up to the Tikhonov term
    H, A = conn._transfer_function, conn._MVAR_Fourier_coefficients
    np.testing.assert_allclose(H @ A, np.broadcast_to(np.eye(3), H.shape), atol=1e-6)


def test_granger_nonnegative_with_nan_diagonal(conn):
    g = conn.pairwise_spectral_granger_prediction()
    assert np.isnan(np.diagonal(g, axis1=-1, axis2=-2)).all()
    v = _offdiag(g)
    assert (v[~np.isnan(v)] > 0).all()                                    # non-positive values are NaN by definition


def test_global_and_canonical_coherence_ranges(conn):
    vals, vecs = conn.global_coherence(max_rank=2)
    assert (vals >= 0).all() and np.isfinite(vecs).all()
    np.testing.assert_allclose(np.linalg.norm(vecs, axis=-2), 1.0, atol=1e-9)
    # the eigenvalues of the CSM sum to its trace = total power
    allv, _ = conn.global_coherence(max_rank=3)
    W, N = allv.shape[:2]
    p = conn.power()                                                      # non-negative bins
    np.testing.assert_allclose(allv[:, : N // 2 + 1].sum(axis=-1), p.sum(axis=-1), rtol=2e-5)
    cc, labels = conn.canonical_coherence(np.array([0, 0, 1]))
    v = cc[..., 0, 1]
    assert (v >= -1e-9).all() and (v <= 1 + 1e-6).all() and list(labels) == [0, 1]


def test_zero_and_tiny_power_channels_do_not_produce_infinities():
    import spectral_connectivity_amd as sc
    rng = np.random.default_rng(0)
    x = rng.standard_normal((128, 4, 3))
    x[..., 1] = 0.0
    x[..., 2] *= 1e-18
    c = sc.Connectivity.from_multitaper(sc.Multitaper(x, sampling_frequency=128.0, time_halfbandwidth_product=2))
    for name in ("coherence_magnitude", "imaginary_coherence", "weighted_phase_lag_index", "phase_lag_index"):
        v = getattr(c, name)()
        assert not np.isinf(v).any()
        fin = v[np.isfinite(v)]
        assert (np.abs(fin) <= 1 + 1e-6).all()


def test_spectrogram_peak_follows_the_signal():
    """A tone that jumps from 40 Hz to 90 Hz half way: every window's power peak sits at the tone of its time span, and
    the window times are the window starts (the qualitative checks of reference tests/test_connectivity.py:735-797)."""
    import spectral_connectivity_amd as sc
    fs, T = 500.0, 2000
    t = np.arange(T) / fs
    f_inst = np.where(t < 2.0, 40.0, 90.0)
    x = np.sin(2 * np.pi * np.cumsum(f_inst) / fs)[:, None, None] + 0.1 * np.random.default_rng(0).standard_normal((T, 3, 1))
    m = sc.Multitaper(x, sampling_frequency=fs, time_halfbandwidth_product=2, time_window_duration=0.5,
                      time_window_step=0.5)
    c = sc.Connectivity.from_multitaper(m)
    p = c.power()[..., 0]
    peak = c.frequencies[np.argmax(p, axis=1)]
    np.testing.assert_allclose(m.time, np.arange(8) * 0.5)
    np.testing.assert_allclose(peak[:4], 40.0, atol=2.0)
    np.testing.assert_allclose(peak[4:], 90.0, atol=2.0)


@pytest.mark.parametrize("L", [64, 65, 100, 101])
def test_nyquist_and_odd_lengths(L):
    """Even and odd window lengths: n_fft // 2 + 1 non-negative frequencies, the even-length Nyquist bin reported
    positive, finite measures in every bin (reference tests/test_connectivity.py:616-732)."""
    import spectral_connectivity_amd as sc
    x = np.random.default_rng(L).standard_normal((L, 6, 2))
    m = sc.Multitaper(x, sampling_frequency=float(L), time_halfbandwidth_product=2)
    c = sc.Connectivity.from_multitaper(m)
    n_fft = m.n_fft_samples
    assert c.frequencies.shape == (n_fft // 2 + 1,) and (c.frequencies >= 0).all()
    if n_fft % 2 == 0:
        assert c.frequencies[-1] == pytest.approx(L / 2)
    coh = c.coherence_magnitude()
    assert coh.shape == (1, n_fft // 2 + 1, 2, 2) and np.isfinite(coh[..., 0, 1]).all()
    assert np.isfinite(c.power()).all() and (c.power() > 0).all()
