"""Pin the CPU oracle (oracle/spectral_oracle.py) against golden vectors produced by the
REAL reference (oracle/gen_golden.py) and against the analytic known-answer vectors of
the reference's own tests (reference tests/test_connectivity.py:25-264)."""
import numpy as np
import pytest

from oracle import spectral_oracle as so

RT = dict(rtol=1e-9, atol=1e-12)


def close(a, b, **kw):
    kw = {**RT, **kw}
    np.testing.assert_allclose(a, b, equal_nan=True, **kw)


def test_f1_cfg1(golden):
    g = golden("f1_cfg1")
    coef, info = so.multitaper_fft(g["x"], fs=float(g["fs"]), NW=float(g["NW"]))
    close(info["tapers"], g["tapers"], rtol=1e-9, atol=1e-11)
    close(coef, g["fft"], atol=1e-12)
    close(info["frequencies"], g["frequencies"])
    close(info["time"], g["time"])
    close(so.nonneg_frequencies(info["frequencies"]), g["conn_frequencies"])
    close(so.power(coef), g["power"])
    close(so.coherency(coef), g["coherency"], atol=1e-10)
    close(so.coherence_magnitude(coef), g["coherence_magnitude"], atol=1e-10)


@pytest.mark.parametrize("det", ["constant", "linear", None])
def test_f2_detrend(golden, det):
    g = golden("f2_detrend")
    coef, _ = so.multitaper_fft(g["x"], fs=float(g["fs"]), NW=float(g["NW"]), detrend_type=det)
    close(coef, g[f"fft_{det}"], atol=1e-11)


@pytest.mark.parametrize("et", list(so.EXPECTATION_AXES))
def test_f3_all_measures(golden, et):
    g = golden("f3_windows_all_measures")
    coef, info = so.multitaper_fft(g["x"], fs=float(g["fs"]), NW=float(g["NW"]),
                                   n_time_samples_per_window=int(g["L"]),
                                   n_time_samples_per_step=int(g["step"]))
    close(coef, g["fft"], atol=1e-12)
    close(info["time"], g["time"])
    for name, fn in so.MEASURES.items():
        close(fn(coef, et), g[f"{et}__{name}"], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("tag,kw", [
    ("L250", dict(n_time_samples_per_window=250)),
    ("L250_N300", dict(n_time_samples_per_window=250, n_fft_samples=300)),
    ("L255", dict(n_time_samples_per_window=255)),
    ("L256_N255", dict(n_time_samples_per_window=256, n_fft_samples=255)),
    ("dur_step", dict(time_window_duration=0.8, time_window_step=0.29)),
])
def test_f4_lengths(golden, tag, kw):
    g = golden("f4_lengths")
    coef, info = so.multitaper_fft(g["x"], fs=float(g["fs"]), NW=float(g["NW"]), **kw)
    close(coef, g[f"{tag}__fft"], atol=1e-11)
    close(info["time"], g[f"{tag}__time"])
    close(info["frequencies"], g[f"{tag}__frequencies"])
    close(so.nonneg_frequencies(info["frequencies"]), g[f"{tag}__conn_frequencies"])
    close(so.coherence_magnitude(coef), g[f"{tag}__coherence_magnitude"], atol=1e-10)
    close(so.power(coef), g[f"{tag}__power"])


@pytest.mark.parametrize("tag,kw", [
    ("ding2", dict(NW=1)),
    ("bacc3", dict(NW=2, n_time_samples_per_window=250)),
])
def test_f5_granger(golden, tag, kw):
    g = golden("f5_granger")
    coef, _ = so.multitaper_fft(g[f"{tag}__x"], fs=200.0, **kw)
    csm = so.expectation_csm_gemm(coef)
    close(csm, g[f"{tag}__csm"], rtol=1e-9, atol=1e-12)
    close(so.expectation_csm_faithful(coef), g[f"{tag}__csm"], rtol=1e-9, atol=1e-12)
    G = so.minimum_phase_decomposition(csm[..., :2, :2])
    close(G, g[f"{tag}__wilson01"], rtol=1e-7, atol=1e-10)
    close(so.pairwise_spectral_granger_prediction(coef), g[f"{tag}__granger"], rtol=1e-6, atol=1e-9)


def test_f13_canonical_coherence_few_observations(golden):
    g = golden("f13_canonical_few_obs")
    coef, _ = so.multitaper_fft(g["x"], fs=float(g["fs"]), NW=float(g["NW"]))
    for tag in ("a", "b"):
        cc, _ = so.canonical_coherence(coef, g[f"labels_{tag}"])
        close(cc, g[f"cc_{tag}"], rtol=1e-9, atol=1e-12)


def test_f12_cholesky_failure_random_restart(golden):
    """A window without a Cholesky factor: the oracle restates the reference's random restart (minimum_phase_
    decomposition.py:78-93) draw for draw, so with the same np.random.seed it lands on the reference's numbers."""
    g = golden("f12_cholesky_fallback")
    coef, _ = so.multitaper_fft(g["x"], fs=float(g["fs"]), NW=float(g["NW"]), n_time_samples_per_window=int(g["L"]))
    for seed in (0, 1):
        np.random.seed(seed)
        with np.errstate(all="ignore"):
            got = so.pairwise_spectral_granger_prediction(coef)
        ref = g[f"granger_seed{seed}"]
        assert np.array_equal(np.isnan(got[0]), np.isnan(ref[0]))
        close(got[0], ref[0], rtol=1e-6, atol=1e-9)                   # the good window, every pair
        close(got[1][:, :2, :2], ref[1][:, :2, :2], rtol=1e-6, atol=1e-9)
        sil = np.stack([got[1][:, 0, 2], got[1][:, 2, 0], got[1][:, 1, 2], got[1][:, 2, 1]])
        assert np.all(np.isnan(sil) | (np.abs(sil) < 1e-9))           # the silent channel's window: no information


@pytest.mark.parametrize("tag", ["var3", "var5"])
def test_f9_mvar_measures(golden, tag):
    """Full C x C Wilson factor and the directed MVAR measures against the real reference."""
    g = golden("f9_mvar")
    coef, _ = so.multitaper_fft(g[f"{tag}__x"], fs=128.0, NW=2, n_time_samples_per_window=256)
    close(so.expectation_csm_gemm(coef), g[f"{tag}__csm"], atol=1e-12)
    q = so.mvar_quantities(coef)
    close(q["G"], g[f"{tag}__minimum_phase_factor"], rtol=1e-7, atol=1e-9)
    close(q["H"], g[f"{tag}__transfer_function"], rtol=1e-7, atol=1e-9)
    close(q["noise_covariance"], g[f"{tag}__noise_covariance"], rtol=1e-7, atol=1e-9)
    close(q["A"], g[f"{tag}__mvar_coefficients"], rtol=1e-6, atol=1e-8)
    for name, fn in so.MVAR_MEASURES.items():
        close(fn(coef, q=q), g[f"{tag}__{name}"], rtol=1e-6, atol=1e-9)


def _same_up_to_phase(u, v, tol):
    """Columns of u and v (..., C, k) span the same lines: |<u_k, v_k>| = |u_k| |v_k|."""
    ip = np.abs(np.sum(np.conj(u) * v, axis=-2))
    nu, nv = np.linalg.norm(u, axis=-2), np.linalg.norm(v, axis=-2)
    np.testing.assert_allclose(ip, nu * nv, rtol=0, atol=tol)


@pytest.mark.parametrize("rank", [1, 2, 4, 5])
def test_f10_global_coherence(golden, rank):
    g = golden("f10_global")
    coef, _ = so.multitaper_fft(g["x"], fs=256.0, NW=2, n_time_samples_per_window=128)
    vals, vecs = so.global_coherence(coef, max_rank=rank)
    close(vals, g[f"rank{rank}__values"], rtol=1e-8, atol=1e-12)
    _same_up_to_phase(vecs, g[f"rank{rank}__vectors"], 1e-6)


def test_f6_canonical(golden):
    g = golden("f6_canonical")
    coef, _ = so.multitaper_fft(g["x"], fs=float(g["fs"]), NW=float(g["NW"]),
                                n_time_samples_per_window=int(g["L"]))
    cc, labels = so.canonical_coherence(coef, g["group_labels"])
    close(cc, g["canonical_coherence"], rtol=1e-8, atol=1e-10)
    assert np.array_equal(labels, g["labels"])


def test_f7_edges(golden):
    g = golden("f7_edges")
    coef, _ = so.multitaper_fft(g["zero__x"], fs=100.0, NW=2)
    for name in ("coherence_magnitude", "imaginary_coherence", "weighted_phase_lag_index",
                 "phase_lag_index"):
        close(so.MEASURES[name](coef), g[f"zero__{name}"], atol=1e-10)
    coef, info = so.multitaper_fft(g["nw175__x"], fs=100.0, NW=1.75)
    close(info["tapers"], g["nw175__tapers"], atol=1e-10)
    close(coef, g["nw175__fft"], atol=1e-11)
    tapers, _ = so.dpss_tapers(128, 1.0, 1, 100.0)
    close(tapers, g["nw1__tapers"], atol=1e-10)
    coef, _ = so.multitaper_fft(g["nw175__x"], fs=100.0, tapers=g["user__tapers"])
    close(coef, g["user__fft"], atol=1e-11)
    # raw coefficients; the reference ran this one with dtype=complex64 matmul
    close(so.expectation_csm_faithful(g["raw__coef"]), g["raw__csm"], rtol=2e-6, atol=2e-6)
    close(so.coherence_magnitude(g["raw__coef"]), g["raw__coherence_magnitude"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("L,NW", [(1024, 3.0), (256, 4.0), (4096, 3.0), (250, 2.0), (64, 2.5)])
def test_f8_dpss(golden, L, NW):
    g = golden("f8_dpss")
    K = int(np.floor(2 * NW - 1))
    tapers, eig = so.dpss_tapers(L, NW, K, fs=1.0, is_low_bias=False)
    close(tapers.T, g[f"L{L}_NW{NW}__tapers"], rtol=1e-7, atol=1e-10)
    close(eig, g[f"L{L}_NW{NW}__eig"], rtol=1e-8, atol=1e-10)


# ---- analytic known answers from the reference's own unit tests -------------------
def _two_signal_coef(phase_a, phase_b, amp_a=2.0, amp_b=3.0):
    coef = np.zeros((1, 1, 1, 1, 2), dtype=complex)
    coef[..., 0] = amp_a * np.exp(1j * phase_a)
    coef[..., 1] = amp_b * np.exp(1j * phase_b)
    return coef


def test_kat_cross_spectrum_and_power():
    # reference tests/test_connectivity.py:25-56 and :82-99
    coef = _two_signal_coef(np.pi / 2, -np.pi / 2)
    csm = so.expectation_csm_faithful(coef)
    close(csm[0, 0], np.array([[4, -6], [-6, 9]], dtype=complex), atol=1e-12)
    close(so.power_two_sided(coef)[0, 0], [4.0, 9.0])


def test_kat_coherency_pli_wpli():
    # reference tests/test_connectivity.py:137-264: |coherency| = 1, phase = pi, NaN diagonal;
    # PLI[0,1] = +1 when signal 0 leads signal 1.
    coef = _two_signal_coef(np.pi / 2, -np.pi / 2)
    coh = so.coherency(coef)[0, 0]
    assert np.isnan(coh[0, 0]) and np.isnan(coh[1, 1])
    close(np.abs(coh[0, 1]), 1.0)
    close(np.abs(np.angle(coh[0, 1])), np.pi)
    coef = _two_signal_coef(np.pi / 2, 0.0)
    close(so.phase_lag_index(coef)[0, 0], [[0, 1], [-1, 0]], atol=1e-15)
    close(so.weighted_phase_lag_index(coef)[0, 0], [[0, 1], [-1, 0]], atol=1e-15)
    close(so.imaginary_coherence(_two_signal_coef(0.3, 0.3))[0, 0, 0, 1], 0.0, atol=1e-15)
    close(so.phase_locking_value(coef)[0, 0, 0, 1], 1.0)


def test_f14_complex_valued_time_series(golden):
    g = golden("f14_complex_series")
    kw = dict(fs=float(g["fs"]), NW=float(g["NW"]), n_time_samples_per_window=int(g["L"]), n_time_samples_per_step=int(g["step"]))
    for det in ("constant", "linear", None):
        coef, _ = so.multitaper_fft(g["x"], detrend_type=det, **kw)
        close(coef, g[f"fft_{det}"], rtol=1e-10, atol=1e-12)
    close(so.coherence_magnitude(coef), g["coherence_magnitude"], atol=1e-10)
    close(so.weighted_phase_lag_index(coef), g["weighted_phase_lag_index"], atol=1e-10)
