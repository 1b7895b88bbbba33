"""The seeded shape sweep of tests/test_gpu_fuzz.py through the float64 engine (`dtype=complex128`, the reference's
default): the same random shapes -- odd channel counts, windows shorter than the FFT length, single trial / taper, every
detrend mode and expectation type -- at float64 tolerances, plus the directed measures on random VAR systems."""
import numpy as np
import pytest

from oracle import spectral_oracle as so
from test_gpu_fuzz import _cases, _var_data

pytestmark = pytest.mark.gpu
SC_PRECISION = "dtype"


def _close(a, b, tol, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, f"{what}: {a.shape} vs {b.shape}"
    assert np.array_equal(np.isnan(a), np.isnan(b)), f"{what}: NaN pattern"
    ok = ~np.isnan(b)
    if ok.any():
        scale = max(np.abs(b[ok]).max(), 1e-300)
        err = np.abs(a[ok] - b[ok]).max()
        assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("cfg", _cases())
def test_random_shapes_against_oracle_float64(cfg):
    import spectral_connectivity_amd as sc
    rng = np.random.default_rng(cfg["seed"])
    x = rng.standard_normal((cfg["T"], cfg["R"], cfg["C"]))
    x += 0.7 * rng.standard_normal((cfg["T"], cfg["R"], 1))
    x += 3.0 + np.linspace(0, 2, cfg["T"])[:, None, None]                   # offset and trend stay in: float64 does not mind
    kw = dict(sampling_frequency=250.0, time_halfbandwidth_product=cfg["NW"], detrend_type=cfg["det"],
              n_time_samples_per_window=cfg["L"], n_time_samples_per_step=cfg["step"])
    m = sc.Multitaper(x, **kw)
    coef, _ = so.multitaper_fft(x, fs=250.0, NW=cfg["NW"], detrend_type=cfg["det"],
                                n_time_samples_per_window=cfg["L"], n_time_samples_per_step=cfg["step"])
    _close(m.fft(), coef, 1e-11, "fft")
    c = sc.Connectivity.from_multitaper(m, expectation_type=cfg["et"])
    n_obs = so.n_observations(coef, cfg["et"])
    _close(c.power(), so.power(coef, cfg["et"]), 1e-10, "power")
    if cfg["C"] < 2:
        return
    # conditioning: how far a measure moves under a 1e-13 relative perturbation of the coefficients (ratios over one or
    # two observations in weak bins amplify rounding); the device is allowed a fixed multiple of that
    pert = coef * (1.0 + 1e-13 * np.random.default_rng(1).standard_normal(coef.shape))
    for name, base in (("coherency", 1e-9), ("coherence_magnitude", 1e-9), ("imaginary_coherence", 1e-9),
                       ("phase_locking_value", 1e-9), ("pairwise_phase_consistency", 1e-9),
                       ("weighted_phase_lag_index", 1e-8), ("debiased_squared_weighted_phase_lag_index", 1e-7)):
        if n_obs < 2 and (name.endswith("phase_lag_index") or name == "pairwise_phase_consistency"):
            continue          # one observation: wPLI is +-1 on the sign of a rounding error, PPC is (1 - 1) / (1 - 1)
        ref = getattr(so, name)(coef, cfg["et"])
        other = getattr(so, name)(pert, cfg["et"])
        both = np.isfinite(ref) & np.isfinite(other)
        sens = np.abs(other[both] - ref[both]).max() / max(np.abs(ref[both]).max(), 1e-300) if both.any() else 0.0
        _close(getattr(c, name)(), ref, base + 100 * sens, name)
    got, ref = c.phase_lag_index(), so.phase_lag_index(coef, cfg["et"])
    assert got.shape == ref.shape and (np.abs(got - ref) > 1e-12).mean() < 1e-3     # sign(Im s) at |Im s| ~ 1e-17 may flip


@pytest.mark.parametrize("C,T,R,L,seed", [(2, 256, 6, None, 1), (3, 200, 8, 100, 2), (4, 256, 5, 128, 3), (6, 128, 10, None, 4),
                                          (9, 256, 12, 128, 5), (20, 256, 16, 128, 6)])
def test_random_var_systems_directed_measures_float64(C, T, R, L, seed):
    import spectral_connectivity_amd as sc
    rng = np.random.default_rng(seed)
    x = _var_data(rng, T, R, C)
    kw = dict(sampling_frequency=200.0, time_halfbandwidth_product=2)
    if L:
        kw["n_time_samples_per_window"] = L
    c = sc.Connectivity.from_multitaper(sc.Multitaper(x, **kw))
    coef, _ = so.multitaper_fft(x, fs=200.0, NW=2, n_time_samples_per_window=L)
    _close(c.pairwise_spectral_granger_prediction(), so.pairwise_spectral_granger_prediction(coef), 1e-7, "granger")
    q = so.mvar_quantities(coef)
    _close(c._minimum_phase_factor, q["G"], 1e-6, "wilson factor")
    for name, fn in so.MVAR_MEASURES.items():
        _close(getattr(c, name)(), fn(coef, q=q), 1e-5, name)
    rank = min(2, C)
    vals, _ = c.global_coherence(max_rank=rank)
    _close(vals, so.global_coherence(coef, max_rank=rank)[0], 1e-8, "global coherence")
    if C >= 3:
        labels = np.arange(C) % 2
        got, lab = c.canonical_coherence(labels)
        ref, _ = so.canonical_coherence(coef, labels)
        _close(got, ref, 1e-7, "canonical coherence")
