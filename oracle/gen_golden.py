"""Generate tests/golden/*.npz from the REAL reference (build container only).

The reference (pure Python) is imported from /root/reference through a namespace shim
(its __init__ needs xarray + package metadata, both absent here).
Only inputs and reference OUTPUTS are stored -- never reference source.  Run:

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz

The GPU box has no /root/reference: tests only read the committed .npz files.
"""
import os
import sys
import types
import warnings

import numpy as np

REF = "/root/reference/spectral_connectivity"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    pkg = types.ModuleType("spectral_connectivity")
    pkg.__path__ = [REF]
    sys.modules["spectral_connectivity"] = pkg
    from spectral_connectivity import connectivity, minimum_phase_decomposition, simulate, transforms
    return transforms, connectivity, minimum_phase_decomposition, simulate


MEASURES = [
    "power", "coherency", "coherence_magnitude", "coherence_phase", "imaginary_coherence",
    "phase_locking_value", "phase_lag_index", "weighted_phase_lag_index",
    "debiased_squared_phase_lag_index", "debiased_squared_weighted_phase_lag_index",
    "pairwise_phase_consistency",
]


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, keys={len(arrays)}")


def gen_f9_mvar(T, Cn, mpd, sim):
    """F9: full C x C Wilson factor and the MVAR measures (DTF, directed coherence, PDC, gPDC, dDTF) on a
    3-channel and a 5-channel VAR system, two time windows each."""
    Multitaper, Connectivity = T.Multitaper, Cn.Connectivity

    def simulate(coefs, cov, n_time, n_trials, seed):
        rs = np.random.default_rng(seed)
        return sim.simulate_MVAR(coefs, noise_covariance=cov, n_time_samples=n_time,
                                 n_trials=n_trials, n_burnin_samples=200, random_state=rs)

    coefs3 = np.zeros((2, 3, 3))
    coefs3[0] = [[0.5, 0.3, 0.4], [-0.5, 0.3, 1.0], [0.0, -0.3, -0.2]]
    coefs3[1] = [[-0.2, 0.0, 0.0], [0.3, -0.3, 0.0], [0.0, 0.0, 0.3]]
    x3 = simulate(coefs3 * 0.6, np.diag([1.0, 0.5, 2.0]), 512, 12, 16)
    # 5-channel system in the style of Baccala & Sameshima (2001), example 3
    r2 = np.sqrt(2.0)
    coefs5 = np.zeros((3, 5, 5))
    coefs5[0, 0, 0] = 0.95 * r2
    coefs5[1, 0, 0] = -0.9025
    coefs5[1, 1, 0] = 0.5
    coefs5[2, 2, 0] = -0.4
    coefs5[1, 3, 0] = -0.5
    coefs5[0, 3, 3] = 0.25 * r2
    coefs5[0, 3, 4] = 0.25 * r2
    coefs5[0, 4, 3] = -0.25 * r2
    coefs5[0, 4, 4] = 0.25 * r2
    x5 = simulate(coefs5, np.eye(5), 512, 10, 17)
    arrs = {}
    for tag, xs in (("var3", x3), ("var5", x5)):
        m = Multitaper(xs, sampling_frequency=128.0, time_halfbandwidth_product=2, n_time_samples_per_window=256)
        c = Connectivity.from_multitaper(m)
        arrs[f"{tag}__x"] = xs
        arrs[f"{tag}__csm"] = c._expectation_cross_spectral_matrix()
        arrs[f"{tag}__minimum_phase_factor"] = c._minimum_phase_factor
        arrs[f"{tag}__transfer_function"] = c._transfer_function
        arrs[f"{tag}__noise_covariance"] = c._noise_covariance
        arrs[f"{tag}__mvar_coefficients"] = c._MVAR_Fourier_coefficients
        for name in ("directed_transfer_function", "directed_coherence", "partial_directed_coherence",
                     "generalized_partial_directed_coherence", "direct_directed_transfer_function"):
            arrs[f"{tag}__{name}"] = getattr(c, name)()
    save("f9_mvar", **arrs)


def gen_f10_global(T, Cn, mpd, sim):
    """F10: global coherence (leading squared singular values / vectors per window and two-sided bin)."""
    rng = np.random.default_rng(10)
    t = np.arange(256) / 256.0
    common = np.sin(2 * np.pi * 40 * t)
    x = 0.6 * rng.standard_normal((256, 6, 5))
    x += common[:, None, None] * np.array([1.0, 0.8, -0.6, 0.3, 0.0])[None, None, :]
    m = T.Multitaper(x, sampling_frequency=256.0, time_halfbandwidth_product=2, n_time_samples_per_window=128)
    c = Cn.Connectivity.from_multitaper(m)
    arrs = dict(x=x)
    for rank in (1, 2, 4, 5):
        v, u = c.global_coherence(max_rank=rank)
        arrs[f"rank{rank}__values"], arrs[f"rank{rank}__vectors"] = v, u
    save("f10_global", **arrs)


def gen_f11_post(T, Cn, mpd, sim):
    """F11: band post-processing of the coherency (phase slope index, delay, group delay) and the
    statistics helpers.  Channel 1 is channel 0 delayed by 5 samples (10 ms at 500 Hz)."""
    import importlib
    st = importlib.import_module("spectral_connectivity.statistics")
    rng = np.random.default_rng(0)
    x = rng.standard_normal((500, 20, 3))
    src = rng.standard_normal((505, 20))
    x[:, :, 0] += 2 * src[5:]
    x[:, :, 1] += 2 * src[:-5]
    m = T.Multitaper(x, sampling_frequency=500.0, time_halfbandwidth_product=3)
    c = Cn.Connectivity.from_multitaper(m)
    res = float(m.frequency_resolution)
    arrs = dict(x=x, coherency=c.coherency(), frequencies=c.frequencies, n_observations=c.n_observations,
                frequency_resolution=res)
    arrs["psi_all"] = c.phase_slope_index()
    arrs["psi_band"] = c.phase_slope_index(frequencies_of_interest=[10, 200])
    arrs["psi_band_res"] = c.phase_slope_index(frequencies_of_interest=[10, 200], frequency_resolution=res)
    arrs["delay_band"] = c.delay(frequencies_of_interest=[10, 200], n_range=2)
    d, sl, r = c.group_delay(frequencies_of_interest=[10, 200], frequency_resolution=res)
    arrs["group_delay"], arrs["group_slope"], arrs["group_r"] = d, sl, r
    # statistics helpers
    p = rng.uniform(size=(6, 7)) ** 3
    arrs["stat_p"] = p
    arrs["stat_bh"] = st.Benjamini_Hochberg_procedure(p, alpha=0.05)
    arrs["stat_bh_none"] = st.Benjamini_Hochberg_procedure(0.5 + 0.5 * p, alpha=0.01)
    arrs["stat_bonf"] = st.Bonferroni_correction(p, alpha=0.05)
    coh1 = 0.9 * rng.uniform(size=(5, 4)) * np.exp(1j * rng.uniform(0, 6, size=(5, 4)))
    coh2 = 0.9 * rng.uniform(size=(5, 4))
    arrs["stat_coh1"], arrs["stat_coh2"] = coh1, coh2
    arrs["stat_fisher2"] = st.coherence_fisher_z_transform(coh1, 40, coh2, 25)
    arrs["stat_pvals"] = st.get_normal_distribution_p_values(arrs["stat_fisher2"])
    arrs["stat_coh_bias"] = st.coherence_bias(40)
    arrs["stat_rate_adj"] = st.coherence_rate_adjustment(10.0, 14.0, np.linspace(0.5, 3, 6), homogeneous_poisson_noise=0.2, dt=0.5)
    lo, hi = st.power_confidence_intervals(7, power=np.linspace(1, 4, 5), ci=0.9)
    arrs["stat_ci_lo"], arrs["stat_ci_hi"] = lo, hi
    arrs["stat_power_bias"], arrs["stat_power_var"] = st.power_bias(35), st.power_variance(35)
    arrs["stat_power_z"] = st.power_fisher_z_transform(np.linspace(1, 4, 5), 35, np.linspace(2, 3, 5), 21)
    save("f11_post", **arrs)


def gen_f12_cholesky_fallback(T, Cn, mpd, sim):
    """F12: a window whose lag-0 covariance has no Cholesky factor (a channel silent in the second window).  The
    reference then restarts EVERY window of the affected pairs from a random positive-definite matrix
    (minimum_phase_decomposition.py:78-93): np.random.seed fixes the draw.  Stored: the prediction for two seeds -- what
    the good windows converge to from a non-Cholesky start (seed-to-seed spread ~1e-5: the iteration stops on
    max|dG| < 1e-8, not at the limit) and what the degenerate window gives (NaN, or 1e-15-size noise)."""
    rng = np.random.default_rng(7)
    n_t, R, C = 512, 30, 3
    e = rng.standard_normal((n_t, R, C))
    x = np.zeros_like(e)
    for t in range(2, n_t):
        x[t] = 0.5 * x[t - 1] - 0.3 * x[t - 2] + e[t]
        x[t, :, 1] += 0.4 * x[t - 1, :, 0]
    x[256:, :, 2] = 0.0
    m = T.Multitaper(x, sampling_frequency=200, time_halfbandwidth_product=2, n_time_samples_per_window=256)
    out = {}
    for seed in (0, 1):
        np.random.seed(seed)
        out[f"granger_seed{seed}"] = Cn.Connectivity.from_multitaper(m).pairwise_spectral_granger_prediction()
    save("f12_cholesky_fallback", x=x, fs=200.0, NW=2.0, L=256, **out)


def gen_f13_canonical_few_observations(T, Cn, mpd, sim):
    """F13: canonical coherence with fewer observations than a group has channels (n_trials * n_tapers = 6; groups of 8,
    4 and 2 channels, then 6, 6 and 2): every pair with such a group comes out 1, the others as usual."""
    rng = np.random.default_rng(1)
    x = rng.standard_normal((128, 2, 14))
    x[:, :, 3] += x[:, :, 9]
    m = T.Multitaper(x, sampling_frequency=100, time_halfbandwidth_product=2)
    c = Cn.Connectivity.from_multitaper(m)
    la, lb = np.array([0] * 8 + [1] * 4 + [2] * 2), np.array([0] * 6 + [1] * 6 + [2] * 2)
    save("f13_canonical_few_obs", x=x, fs=100.0, NW=2.0, labels_a=la, labels_b=lb,
         cc_a=c.canonical_coherence(la)[0], cc_b=c.canonical_coherence(lb)[0])


def gen_f14_complex_series(T, Cn, mpd, sim):
    """F14: complex-valued time series (the reference's generic fft takes them, transforms.py:1402-1405): two-sided
    coefficients for every detrend mode, and the measures of the non-negative bins."""
    rng = np.random.default_rng(14)
    x = rng.standard_normal((320, 3, 5)) + 1j * rng.standard_normal((320, 3, 5))
    x[:, :, 1] += 0.6 * np.roll(x[:, :, 0], 2, axis=0)
    x += (np.linspace(0, 1, 320) * (1 + 2j))[:, None, None]
    out = dict(x=x, fs=200.0, NW=2.0, L=128, step=64)
    for det in ("constant", "linear", None):
        m = T.Multitaper(x, sampling_frequency=200.0, time_halfbandwidth_product=2, n_time_samples_per_window=128,
                         n_time_samples_per_step=64, detrend_type=det)
        out[f"fft_{det}"] = m.fft()
    c = Cn.Connectivity.from_multitaper(m)       # detrend None: the last one
    for name in ("power", "coherency", "coherence_magnitude", "weighted_phase_lag_index", "phase_locking_value",
                 "pairwise_spectral_granger_prediction"):
        out[name] = getattr(c, name)()
    save("f14_complex_series", **out)


def gen_api_surface(*_):
    """Public names of the reference (functions, classes, methods, properties) with their argument names and default
    values, as data: tests/golden/api_surface.json.  The drop-in mirrors exactly this surface."""
    import ast
    import json
    root = "/root/reference/spectral_connectivity"

    def describe(fn):
        a = fn.args
        pos = a.posonlyargs + a.args
        dflt = [None] * (len(pos) - len(a.defaults)) + [ast.unparse(x) for x in a.defaults]
        args = [[p.arg, d] for p, d in zip(pos, dflt)]
        args += [[k.arg, ast.unparse(v) if v is not None else None] for k, v in zip(a.kwonlyargs, a.kw_defaults)]
        return {"args": args, "vararg": bool(a.vararg), "kwarg": bool(a.kwarg)}

    surface = {}
    for f in sorted(os.listdir(root)):
        if not f.endswith(".py") or f == "__init__.py":
            continue
        tree = ast.parse(open(os.path.join(root, f)).read())
        mod = {}
        for n in tree.body:
            if isinstance(n, ast.FunctionDef) and not n.name.startswith("_"):
                mod[n.name] = describe(n)
            elif isinstance(n, ast.ClassDef) and not n.name.startswith("_"):
                for m in n.body:
                    if isinstance(m, ast.FunctionDef) and (not m.name.startswith("_") or m.name == "__init__"):
                        mod[n.name + "." + m.name] = describe(m)
            elif isinstance(n, ast.Assign):
                for tg in n.targets:
                    if isinstance(tg, ast.Name) and tg.id.isupper():
                        mod[tg.id] = {"constant": ast.unparse(n.value)}
        surface[f[:-3]] = mod
    init = ast.parse(open(os.path.join(root, "__init__.py")).read())
    for n in init.body:
        if isinstance(n, ast.Assign) and any(isinstance(t, ast.Name) and t.id == "__all__" for t in n.targets):
            surface["__all__"] = sorted(ast.literal_eval(n.value))
    with open(os.path.join(OUT, "api_surface.json"), "w") as fh:
        json.dump(surface, fh, indent=1, sort_keys=True)
    print("api_surface:", sum(len(v) for k, v in surface.items() if k != "__all__"), "names")


def main():
    os.makedirs(OUT, exist_ok=True)
    warnings.simplefilter("ignore")
    T, Cn, mpd, sim = import_reference()
    if len(sys.argv) > 1:                      # python oracle/gen_golden.py f9 : only the named fixtures
        for name in sys.argv[1:]:
            {"f9": gen_f9_mvar, "f10": gen_f10_global, "f11": gen_f11_post, "f12": gen_f12_cholesky_fallback,
             "f13": gen_f13_canonical_few_observations, "f14": gen_f14_complex_series, "api": gen_api_surface}[name](T, Cn, mpd, sim)
        return
    Multitaper, Connectivity = T.Multitaper, Cn.Connectivity

    # F1: BASELINE cfg1 -- 2 ch x 1 trial x 1024, sine + noise, NW=3, single window
    fs = 1000.0
    t = np.arange(1024) / fs
    rng = np.random.default_rng(0)
    x = np.stack([np.sin(2 * np.pi * 50 * t) + 0.5 * rng.standard_normal(1024),
                  np.sin(2 * np.pi * 50 * t + np.pi / 4) + 0.5 * rng.standard_normal(1024)],
                 axis=-1)[:, None, :]
    m = Multitaper(x, sampling_frequency=fs, time_halfbandwidth_product=3)
    c = Connectivity.from_multitaper(m)
    save("f1_cfg1", x=x, fs=fs, NW=3.0, tapers=m.tapers, fft=m.fft(),
         frequencies=m.frequencies, time=m.time, conn_frequencies=c.frequencies,
         power=c.power(), coherency=c.coherency(), coherence_magnitude=c.coherence_magnitude())

    # F2: detrend variants
    x = np.random.default_rng(1).standard_normal((256, 6, 5))
    x += np.linspace(0, 3, 256)[:, None, None] * np.arange(1, 6)[None, None, :]
    arrs = dict(x=x, fs=500.0, NW=3.0)
    for det in ("constant", "linear", None):
        m = Multitaper(x, sampling_frequency=500.0, time_halfbandwidth_product=3, detrend_type=det)
        arrs[f"fft_{det}"] = m.fft()
    save("f2_detrend", **arrs)

    # F3: sliding windows, every measure, every expectation type (kept small: 3 ch, L=64)
    x = np.random.default_rng(3).standard_normal((256, 3, 3))
    tt = np.arange(256) / fs
    for ch in range(3):
        x[:, :, ch] += 0.8 * np.sin(2 * np.pi * 120 * tt + 2 * np.pi * ch / 3)[:, None]
    m = Multitaper(x, sampling_frequency=fs, time_halfbandwidth_product=2,
                   n_time_samples_per_window=64, n_time_samples_per_step=32)
    arrs = dict(x=x, fs=fs, NW=2.0, L=64, step=32, fft=m.fft(), time=m.time,
                tapers=m.tapers, frequencies=m.frequencies)
    for et in Cn.EXPECTATION:
        c = Connectivity.from_multitaper(m, expectation_type=et)
        for name in MEASURES:
            arrs[f"{et}__{name}"] = getattr(c, name)()
    save("f3_windows_all_measures", **arrs)

    # F4: non power-of-two lengths (Nyquist handling, zero padding, truncation)
    x = np.random.default_rng(4).standard_normal((600, 3, 4))
    arrs = dict(x=x, fs=250.0, NW=2.0)
    for tag, kw in {
        "L250": dict(n_time_samples_per_window=250),
        "L250_N300": dict(n_time_samples_per_window=250, n_fft_samples=300),
        "L255": dict(n_time_samples_per_window=255),
        "L256_N255": dict(n_time_samples_per_window=256, n_fft_samples=255),
        "dur_step": dict(time_window_duration=0.8, time_window_step=0.29),
    }.items():
        m = Multitaper(x, sampling_frequency=250.0, time_halfbandwidth_product=2, **kw)
        c = Connectivity.from_multitaper(m)
        arrs[f"{tag}__fft"] = m.fft()
        arrs[f"{tag}__time"] = m.time
        arrs[f"{tag}__frequencies"] = m.frequencies
        arrs[f"{tag}__conn_frequencies"] = c.frequencies
        arrs[f"{tag}__coherence_magnitude"] = c.coherence_magnitude()
        arrs[f"{tag}__power"] = c.power()
    save("f4_lengths", **arrs)

    # F5: MVAR systems -> CSM, Wilson factor, pairwise spectral Granger
    # Baccala & Sameshima 3-channel-like VAR(2) and Ding 2-channel VAR(2)
    def simulate(coefs, cov, n_time, n_trials, seed):
        rs = np.random.default_rng(seed)
        return sim.simulate_MVAR(coefs, noise_covariance=cov, n_time_samples=n_time,
                                 n_trials=n_trials, n_burnin_samples=200, random_state=rs)

    coefs2 = np.array([[[0.9, 0.0], [0.16, 0.8]], [[-0.5, 0.0], [-0.2, -0.5]]])
    cov2 = np.array([[1.0, 0.4], [0.4, 0.7]])
    x2 = simulate(coefs2, cov2, 1000, 30, 5)
    coefs3 = np.zeros((2, 3, 3))
    coefs3[0] = [[0.5, 0.3, 0.4], [-0.5, 0.3, 1.0], [0.0, -0.3, -0.2]]
    coefs3[1] = [[-0.2, 0.0, 0.0], [0.3, -0.3, 0.0], [0.0, 0.0, 0.3]]
    x3 = simulate(coefs3 * 0.6, np.eye(3), 500, 20, 6)
    arrs = {}
    for tag, xs, kw in (("ding2", x2, dict(time_halfbandwidth_product=1)),
                        ("bacc3", x3, dict(time_halfbandwidth_product=2,
                                           n_time_samples_per_window=250))):
        m = Multitaper(xs, sampling_frequency=200.0, **kw)
        c = Connectivity.from_multitaper(m)
        csm = c._expectation_cross_spectral_matrix()
        arrs[f"{tag}__x"] = xs
        arrs[f"{tag}__csm"] = csm
        arrs[f"{tag}__power"] = c.power()
        arrs[f"{tag}__granger"] = c.pairwise_spectral_granger_prediction()
        sub = csm[..., :2, :2]
        arrs[f"{tag}__wilson01"] = mpd.minimum_phase_decomposition(sub)
    save("f5_granger", **arrs)

    # F6: canonical coherence, 12 channels in 3 groups (2/4/6), 8 trials
    x = np.random.default_rng(6).standard_normal((512, 8, 12))
    labels = np.array([0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2])
    src = np.random.default_rng(66).standard_normal((512, 8))
    x[:, :, :2] += 0.7 * src[..., None]
    x[:, :, 6:9] += 0.7 * src[..., None]
    m = Multitaper(x, sampling_frequency=fs, time_halfbandwidth_product=3,
                   n_time_samples_per_window=256)
    c = Connectivity.from_multitaper(m)
    cc, lab = c.canonical_coherence(labels)
    save("f6_canonical", x=x, fs=fs, NW=3.0, L=256, group_labels=labels,
         canonical_coherence=cc, labels=lab)

    # F7: edge cases
    arrs = {}
    x = np.random.default_rng(7).standard_normal((200, 5, 4))
    x[:, :, 2] = 0.0                                       # zero-power channel
    m = Multitaper(x, sampling_frequency=100.0, time_halfbandwidth_product=2)
    c = Connectivity.from_multitaper(m)
    arrs["zero__x"] = x
    arrs["zero__coherence_magnitude"] = c.coherence_magnitude()
    arrs["zero__imaginary_coherence"] = c.imaginary_coherence()
    arrs["zero__weighted_phase_lag_index"] = c.weighted_phase_lag_index()
    arrs["zero__phase_lag_index"] = c.phase_lag_index()
    # NW=1.75 -> 2 tapers requested; low-bias cut
    x = np.random.default_rng(8).standard_normal((128, 3, 2))
    m = Multitaper(x, sampling_frequency=100.0, time_halfbandwidth_product=1.75)
    arrs["nw175__x"] = x
    arrs["nw175__tapers"] = m.tapers
    arrs["nw175__fft"] = m.fft()
    m = Multitaper(x, sampling_frequency=100.0, time_halfbandwidth_product=1.0)
    arrs["nw1__tapers"] = m.tapers
    # user supplied tapers (Hann, 2 columns)
    user = np.stack([np.hanning(128), np.hanning(128) ** 2], axis=1)
    m = Multitaper(x, sampling_frequency=100.0, tapers=user)
    arrs["user__tapers"] = user
    arrs["user__fft"] = m.fft()
    # synthetic coefficients fed straight to Connectivity (complex64 dtype path)
    rs = np.random.default_rng(9)
    coef = rs.standard_normal((2, 5, 3, 16, 4)) + 1j * rs.standard_normal((2, 5, 3, 16, 4))
    c = Connectivity(coef, dtype=np.complex64)
    arrs["raw__coef"] = coef
    arrs["raw__coherence_magnitude"] = c.coherence_magnitude()
    arrs["raw__csm"] = c._expectation_cross_spectral_matrix()
    save("f7_edges", **arrs)

    # DPSS tapers at the BASELINE shapes (float64, already scaled by sqrt(fs))
    arrs = {}
    for L, NW in ((1024, 3.0), (256, 4.0), (4096, 3.0), (250, 2.0), (64, 2.5)):
        tap, eig = T.dpss_windows(L, NW, int(np.floor(2 * NW - 1)), is_low_bias=False)
        arrs[f"L{L}_NW{NW}__tapers"] = np.asarray(tap)
        arrs[f"L{L}_NW{NW}__eig"] = np.asarray(eig)
    save("f8_dpss", **arrs)
    gen_f9_mvar(T, Cn, mpd, sim)
    gen_f10_global(T, Cn, mpd, sim)
    gen_f11_post(T, Cn, mpd, sim)
    gen_f12_cholesky_fallback(T, Cn, mpd, sim)
    gen_f13_canonical_few_observations(T, Cn, mpd, sim)
    gen_f14_complex_series(T, Cn, mpd, sim)
    gen_api_surface()


if __name__ == "__main__":
    main()
