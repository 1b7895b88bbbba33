"""Seeded sweep over shapes: every fast path (fused FFT kernels, rocFFT path, fused / separate stage B, split bins)
against the oracle on small random problems -- odd channel counts, windows shorter than the FFT length, single
trial, single taper, few observations, every detrend mode and expectation type."""
import numpy as np
import pytest

from oracle import spectral_oracle as so

pytestmark = pytest.mark.gpu


def _cases():
    import os
    rng = np.random.default_rng(2024)
    out = []
    n_cases = int(os.environ.get("SC_FUZZ_CASES", "36"))      # a longer sweep: SC_FUZZ_CASES=400 pytest tests/test_gpu_fuzz.py
    for k in range(n_cases):
        channels = [1, 2, 3, 5, 8, 16, 17, 31, 32, 33, 48, 64, 65, 96, 127, 128]
        if k >= 36:      # the extra cases also walk the kernel-selection boundaries (42 / 48 / 50 channels, odd counts)
            channels += [4, 6, 7, 19, 34, 40, 41, 42, 43, 44, 46, 47, 49, 50, 51, 63, 100, 129, 130, 160, 162, 192, 200, 224, 226, 250, 256]
        C = int(rng.choice(channels))
        L = int(rng.choice([16, 50, 64, 100, 128, 200, 256]))
        step = int(rng.choice([L, max(L // 2, 1), max(L // 3, 1)]))
        W = int(rng.integers(1, 4))
        T = L + (W - 1) * step + int(rng.integers(0, max(step - 1, 1)))
        R = int(rng.choice([1, 2, 3, 7, 20]))
        NW = float(rng.choice([1.0, 2.0, 2.5, 4.0]))
        det = [None, "constant", "linear"][k % 3]
        et = ["trials_tapers", "trials", "tapers", "time", "time_trials", "time_tapers", "time_trials_tapers"][k % 7]
        out.append(pytest.param(dict(C=C, L=L, step=step, T=T, R=R, NW=NW, det=det, et=et, seed=k), id=f"{k}-C{C}-L{L}-R{R}-{et}"))
    return out


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
def test_random_shapes_against_oracle(cfg):
    import spectral_connectivity_amd as sc
    rng = np.random.default_rng(cfg["seed"])
    x = rng.standard_normal((cfg["T"], cfg["R"], cfg["C"]))
    x += 0.7 * rng.standard_normal((cfg["T"], cfg["R"], 1))                 # shared component: non-trivial coherence
    if cfg["det"] is not None:
        # a trend for the detrend modes.  (Left in, it would put ~1e3 x the noise power into the DC bin, and an f32
        # transform leaks 1e-7 of THAT into every other bin: the weak bins then carry 1e-4 relative error, which is
        # a property of single precision on un-detrended trending data, not of a kernel.)
        x += np.linspace(0, 2, cfg["T"])[:, None, None]
    kw = dict(sampling_frequency=250.0, time_halfbandwidth_product=cfg["NW"], detrend_type=cfg["det"],
              n_time_samples_per_window=cfg["L"], n_time_samples_per_step=cfg["step"])
    m = sc.Multitaper(x, **kw)
    coef, info = so.multitaper_fft(x, fs=250.0, NW=cfg["NW"], detrend_type=cfg["det"],
                                   n_time_samples_per_window=cfg["L"], n_time_samples_per_step=cfg["step"])
    _close(m.fft(), coef, 2e-5, "fft")
    c = sc.Connectivity.from_multitaper(m, expectation_type=cfg["et"])
    n_obs = so.n_observations(coef, cfg["et"])
    _close(c.power(), so.power(coef, cfg["et"]), 2e-5, "power")
    if cfg["C"] >= 2:
        # Conditioning: how far the measure moves when the INPUT is merely rounded to f32 (what the device is handed).
        # With one or two observations and weak bins that alone reaches 1e-5..1e-4; the device path (f32 transform, f32
        # products) is allowed a fixed multiple of it on top of the plain f32 tolerance.
        coef32, _ = so.multitaper_fft(x.astype(np.float32).astype(np.float64), fs=250.0, NW=cfg["NW"],
                                      detrend_type=cfg["det"], n_time_samples_per_window=cfg["L"],
                                      n_time_samples_per_step=cfg["step"])

        def check(name, base_tol):
            ref = getattr(so, name)(coef, cfg["et"])
            sens = np.nanmax(np.abs(getattr(so, name)(coef32, cfg["et"]) - ref)) if np.isfinite(ref).any() else 0.0
            _close(getattr(c, name)(), ref, base_tol + 60 * sens, name)

        check("coherence_magnitude", 3e-5)
        check("imaginary_coherence", 3e-5)
        if n_obs >= 2:                       # with one observation wPLI is +-1 and flips on f32 rounding of Im s ~ 0
            check("weighted_phase_lag_index", 1e-4)

def _var_data(rng, T, R, C):
    """Stable random VAR(2) with sparse coupling, driven by white noise of unequal variances."""
    A1 = np.diag(rng.uniform(0.2, 0.6, C))
    A2 = np.diag(rng.uniform(-0.4, -0.1, C))
    for _ in range(C):
        i, j = rng.integers(0, C, 2)
        if i != j:
            A1[i, j] = rng.uniform(-0.35, 0.35)
    rho = max(np.abs(np.linalg.eigvals(np.block([[A1, A2], [np.eye(C), np.zeros((C, C))]]))))
    if rho >= 0.9:
        A1, A2 = A1 * 0.85 / rho, A2 * (0.85 / rho) ** 2
    x = np.zeros((T + 100, R, C))
    e = rng.standard_normal((T + 100, R, C)) * rng.uniform(0.5, 1.5, C)
    for t in range(2, T + 100):
        x[t] = x[t - 1] @ A1.T + x[t - 2] @ A2.T + e[t]
    return x[100:]


@pytest.mark.parametrize("C,T,R,L,seed", [(2, 256, 6, None, 1), (3, 200, 8, 100, 2), (4, 256, 5, 128, 3), (6, 128, 10, None, 4),
                                          (9, 256, 12, 128, 5)])
def test_random_var_systems_directed_measures(C, T, R, L, seed):
    """Pairwise Granger, full Wilson factor, DTF / PDC / gPDC / dDTF / directed coherence, global and canonical
    coherence on random stable VAR(2) systems against the oracle."""
    import spectral_connectivity_amd as sc
    rng = np.random.default_rng(seed)
    x = _var_data(rng, T, R, C)
    kw = dict(sampling_frequency=200.0, time_halfbandwidth_product=2)
    if L:
        kw["n_time_samples_per_window"] = L
    m = sc.Multitaper(x, **kw)
    c = sc.Connectivity.from_multitaper(m)
    coef, _ = so.multitaper_fft(x, fs=200.0, NW=2, n_time_samples_per_window=L)
    _close(c.pairwise_spectral_granger_prediction(), so.pairwise_spectral_granger_prediction(coef), 2e-4, "granger")
    q = so.mvar_quantities(coef)
    _close(c._minimum_phase_factor, q["G"], 2e-4, "wilson factor")
    for name, fn in so.MVAR_MEASURES.items():
        _close(getattr(c, name)(), fn(coef, q=q), 5e-4, name)
    rank = min(2, C)
    vals, _ = c.global_coherence(max_rank=rank)
    _close(vals, so.global_coherence(coef, max_rank=rank)[0], 5e-5, "global coherence")
    if C >= 3:
        labels = np.arange(C) % 2
        got, lab = c.canonical_coherence(labels)
        ref, _ = so.canonical_coherence(coef, labels)
        _close(got, ref, 1e-4, "canonical coherence")


@pytest.mark.parametrize("seed", range(8))
def test_random_group_structures_canonical_coherence(seed):
    """Canonical coherence over shuffled labels with 2-9 groups of 1-16 channels each (the per-bin kernel's whole
    range: single-channel groups, a full 16-channel group, equal and ragged sizes) against the oracle's SVD form."""
    import spectral_connectivity_amd as sc
    rng = np.random.default_rng(1000 + seed)
    G = int(rng.integers(2, 10))
    sizes = rng.integers(1, 17, G)
    if seed % 4 == 0:
        sizes[:] = rng.integers(1, 4)
    if seed % 3 == 0:
        sizes[rng.integers(G)] = 16
    C = int(sizes.sum())
    labels = np.repeat(np.arange(G), sizes)[rng.permutation(C)]
    T = int(rng.choice([64, 100, 128]))
    R = int(rng.integers(max(3, (sizes.max() + 2) // 3 + 1), 12))       # n_obs = 3 R >= largest group
    x = rng.standard_normal((T, R, C)) @ (rng.standard_normal((C, C)) * 0.4 + np.eye(C)).T
    m = sc.Multitaper(x, sampling_frequency=100.0, time_halfbandwidth_product=2)
    got, lab = sc.Connectivity.from_multitaper(m).canonical_coherence(labels)
    coef, _ = so.multitaper_fft(x, fs=100.0, NW=2)
    ref, ref_lab = so.canonical_coherence(coef, labels)
    assert list(lab) == list(ref_lab)
    _close(got, ref, 5e-5, f"canonical coherence, group sizes {list(sizes)}")
