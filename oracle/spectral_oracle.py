"""CPU oracle for the multitaper -> cross-spectral-matrix -> connectivity hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``spectral_connectivity_amd/`` may import
this module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and there only as the checker / timed CPU baseline.

It is a float64 NumPy restatement of the reference algorithm (Eden-Kramer-Lab/
spectral_connectivity @ /root/reference).  Every function cites the reference
``file:line`` it follows.  Parity is PINNED: ``oracle/gen_golden.py`` imports the real
reference in the build container and writes ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this module against those vectors
(rtol 1e-9 or tighter) and against the analytic known-answer vectors of the reference's
own tests (tests/test_connectivity.py:25-264 of the reference).

Third-party arithmetic the reference delegates to (not under /root/reference):
scipy.fft (pocketfft) ``fft/ifft/next_fast_len/fftfreq``, scipy.linalg
``eigvals_banded``/``lstsq``, numpy ``matmul/mean/linalg.solve/cholesky/svd``.  The
reference does not pin versions (pyproject.toml:42-47: numpy>=1.24, scipy>=1.10); this
oracle and the golden vectors were produced with numpy 2.2.6 / scipy 1.15.3.

Layout conventions (same as the reference):
    time series          (T, R, C)            time, trials, signals
    Fourier coefficients (W, R, K, N, C)      windows, trials, tapers, fft bins, signals
"""

from itertools import combinations

import numpy as np
from scipy.fft import fft, fftfreq, ifft, next_fast_len
from scipy.signal.windows import dpss as _scipy_dpss

EPS = np.finfo(np.float64).eps

# connectivity.py:67-75 -- which of the (window, trial, taper) axes are averaged
EXPECTATION_AXES = {
    "time": (0,),
    "trials": (1,),
    "tapers": (2,),
    "time_trials": (0, 1),
    "time_tapers": (0, 2),
    "trials_tapers": (1, 2),
    "time_trials_tapers": (0, 1, 2),
}


# --------------------------------------------------------------------------- transform
def window_geometry(n_time, fs, time_window_duration=None, time_window_step=None,
                    n_time_samples_per_window=None, n_time_samples_per_step=None,
                    n_fft_samples=None):
    """L, step, N, W exactly as the reference derives them.

    transforms.py:996-1022 (L = around(duration*fs), else given, else T),
    transforms.py:1050-1073 (step = int(step_duration*fs): truncation, else given, else L),
    transforms.py:1024-1036 (N = next_fast_len(L) unless given),
    transforms.py:1363-1365 (W = floor(T/step - L/step + 1) in floating point).
    """
    if n_time_samples_per_window is None and time_window_duration is None:
        L = int(n_time)
    elif time_window_duration is not None:
        L = int(np.around(time_window_duration * fs))
    else:
        L = int(n_time_samples_per_window)
    if n_time_samples_per_step is None and time_window_step is None:
        step = L
    elif time_window_step is not None:
        step = int(time_window_step * fs)
    else:
        step = int(n_time_samples_per_step)
    N = int(next_fast_len(L)) if n_fft_samples is None else int(n_fft_samples)
    W = int(np.floor((n_time / step) - (L / step) + 1))
    return L, step, N, W


def sliding_windows(x, L, step):
    """(T, R, C) -> (W, R, C, L): window w covers samples [w*step, w*step+L).

    transforms.py:1311-1374 (as_strided view + copy).
    """
    T = x.shape[0]
    W = int(np.floor((T / step) - (L / step) + 1))
    out = np.empty((W,) + x.shape[1:] + (L,), dtype=x.dtype)
    for w in range(W):
        out[w] = np.moveaxis(x[w * step: w * step + L], 0, -1)
    return out


def detrend(windows, kind):
    """Detrend along the last axis.  transforms.py:1798-1915.

    'constant': subtract the mean (:1859-1860).  'linear': subtract the least-squares
    line fitted on the abscissa (1..L)/L with an intercept (:1903-1909, single segment).
    None: untouched (transforms.py:1164).
    """
    if kind is None:
        return windows
    if kind in ("constant", "c"):
        return windows - windows.mean(axis=-1, keepdims=True)
    if kind in ("linear", "l"):
        L = windows.shape[-1]
        A = np.ones((L, 2))
        A[:, 0] = np.arange(1, L + 1) / L
        # (complex series: the real regressors fit the real and the imaginary part separately, transforms.py:1903-1909)
        flat = windows.reshape(-1, L).T.astype(np.complex128 if np.iscomplexobj(windows) else np.float64)
        coef = np.linalg.lstsq(A, flat, rcond=None)[0]
        return (flat - A @ coef).T.reshape(windows.shape)
    raise ValueError(f"Invalid trend type '{kind}'")


def dpss_tapers(L, NW, n_tapers, fs, is_low_bias=True):
    """DPSS tapers scaled by sqrt(fs), shape (L, K'); and concentration eigenvalues.

    Follows transforms.py:1408-1440 (_make_tapers), :1539-1613 (dpss_windows),
    :1717-1745 (sign convention), :1768-1795 (eigenvalues via autocorrelation),
    :1758-1765 (keep eigenvalue > 0.9, or the arg-max if none).  The eigenvectors
    themselves come from scipy.signal.windows.dpss (same tridiagonal problem,
    LAPACK instead of the reference's inverse iteration; agrees to ~1e-13).
    """
    n_tapers = int(n_tapers)
    tapers = np.atleast_2d(_scipy_dpss(L, NW, n_tapers, sym=True, norm=2))
    tapers = np.array(tapers, dtype=np.float64)
    # symmetric tapers: positive mean
    flip = tapers[::2].sum(axis=1) < 0
    tapers[::2][flip] *= -1
    # antisymmetric tapers: positive slope up to the first (largest) peak of the first half
    peak = np.argmax(np.abs(tapers[1::2, : L // 2]), axis=1)
    for i, p in enumerate(peak):
        if tapers[2 * i + 1, :p].sum() < 0:
            tapers[2 * i + 1] *= -1
    # concentration: autocorrelation of each taper dotted with the ideal low-pass kernel
    half_bw = float(NW) / L
    t = np.arange(L, dtype=np.float64)
    nfft = next_fast_len(2 * L - 1)
    spec = fft(tapers, nfft, axis=-1)
    acorr = np.real(ifft(spec * spec.conj(), axis=-1))[:, :L]
    kernel = 4 * half_bw * np.sinc(2 * half_bw * t)
    kernel[0] = 2 * half_bw
    eig = acorr @ kernel
    if is_low_bias:
        keep = eig > 0.9
        if not keep.any():
            keep = np.zeros_like(keep)
            keep[np.argmax(eig)] = True
        tapers, eig = tapers[keep], eig[keep]
    return tapers.T * np.sqrt(fs), eig


def multitaper_fft(x, fs=1000.0, NW=3, detrend_type="constant", n_tapers=None,
                   tapers=None, is_low_bias=True, **geometry):
    """Time series (T,R,C) -> coefficients (W,R,K,N,C) complex128 (two-sided).

    Multitaper.fft, transforms.py:1147-1171: sliding windows -> detrend -> multiply by
    every taper -> fft(n=N) along the window axis -> / fs (transforms.py:1402-1405).
    Returns (coefficients, info) with info = dict(L, step, N, W, tapers, frequencies, time).
    """
    x = np.asarray(x)
    L, step, N, W = window_geometry(x.shape[0], fs, **geometry)
    if tapers is None:
        if n_tapers is None:
            n_tapers = int(np.floor(2.0 * NW - 1))           # transforms.py:979-994
        tapers, _ = dpss_tapers(L, NW, n_tapers, fs, is_low_bias)
    win = detrend(sliding_windows(x, L, step), detrend_type)  # (W,R,C,L)
    projected = win[..., np.newaxis] * tapers[np.newaxis, np.newaxis, ...]   # (W,R,C,L,K)
    coef = fft(projected, n=N, axis=-2) / fs                                 # (W,R,C,N,K)
    coef = coef.swapaxes(2, -1)                                              # (W,R,K,N,C)
    info = dict(L=L, step=step, N=N, W=W, tapers=tapers,
                frequencies=fftfreq(N, 1.0 / fs),                # transforms.py:1038-1048
                time=np.arange(W) * step / fs)                   # transforms.py:1075-1091
    return coef, info


def nonneg_frequencies(frequencies):
    """connectivity.py:402-424: first N//2+1 bins, Nyquist made positive."""
    n = len(frequencies)
    f = np.array(frequencies[: n // 2 + 1], dtype=np.float64)
    if len(f) and f[-1] < 0:
        f[-1] = abs(f[-1])
    return f


# ------------------------------------------------------------------ expectation / CSM
def _mean(a, expectation_type):
    return a.mean(axis=EXPECTATION_AXES[expectation_type])


def n_observations(coef, expectation_type):
    """connectivity.py:594-610."""
    return int(np.prod([coef.shape[a] for a in EXPECTATION_AXES[expectation_type]]))


def power_two_sided(coef, expectation_type="trials_tapers"):
    """connectivity.py:441-445: E[|X|^2], all N bins."""
    return _mean((coef * coef.conj()).real, expectation_type)


def expectation_csm_faithful(coef, expectation_type="trials_tapers", fcn=None):
    """Op-for-op the reference: per-observation outer product, fcn, then mean.

    connectivity.py:447-461 + :1799-1822 (matmul of (...,C,1) @ conj((...,C,1))^T),
    :463-492 (fcn hook, expectation).  Memory O(W R K N C^2): small inputs only; this
    is also what bench.py times as the single-core CPU baseline.
    """
    a = coef[..., np.newaxis]
    per_obs = np.matmul(a, a.swapaxes(-1, -2).conj())
    if fcn is not None:
        per_obs = fcn(per_obs)
    return _mean(per_obs, expectation_type)


def expectation_csm_gemm(coef, expectation_type="trials_tapers"):
    """Same E[X_i conj X_j] as one contraction over the averaged axes (no per-obs temp).

    Algebraically identical to expectation_csm_faithful(fcn=None);
    used as float64 ground truth at shapes where the faithful form cannot be allocated.
    """
    axes = EXPECTATION_AXES[expectation_type]
    letters = "wrk"
    kept = "".join(ch for i, ch in enumerate(letters) if i not in axes)
    n = n_observations(coef, expectation_type)
    return np.einsum(f"wrkni,wrknj->{kept}nij", coef, coef.conj(), optimize=True) / n


def _zero_diag_imag(x):
    im = np.array(x.imag)
    idx = np.arange(im.shape[-1])
    im[..., idx, idx] = 0.0
    return im


def _take_nonneg(a, axis):
    n = a.shape[axis]
    return np.take(a, np.arange(n // 2 + 1), axis=axis)


def power(coef, expectation_type="trials_tapers"):
    """connectivity.py:612-630."""
    return _take_nonneg(power_two_sided(coef, expectation_type), -2)


def coherency(coef, expectation_type="trials_tapers", csm=None):
    """connectivity.py:632-657: S_ij / max(sqrt(P_i P_j), eps), NaN diagonal."""
    p = power_two_sided(coef, expectation_type)
    norm = np.maximum(np.sqrt(p[..., :, np.newaxis] * p[..., np.newaxis, :]), EPS)
    if csm is None:
        csm = expectation_csm_faithful(coef, expectation_type)
    out = csm / norm
    idx = np.arange(out.shape[-1])
    out[..., idx, idx] = np.nan
    return _take_nonneg(out, -3)


def coherence_magnitude(coef, expectation_type="trials_tapers", csm=None):
    """connectivity.py:675-702: clip(|coherency|^2, 0, 1)."""
    return np.clip(np.abs(coherency(coef, expectation_type, csm)) ** 2, 0, 1)


def coherence_phase(coef, expectation_type="trials_tapers", csm=None):
    """connectivity.py:659-673."""
    return np.angle(coherency(coef, expectation_type, csm))


def imaginary_coherence(coef, expectation_type="trials_tapers", csm=None):
    """connectivity.py:704-743: clip(|Im S| / max(sqrt(PiPj), eps), 0, 1); diagonal 0."""
    p = power_two_sided(coef, expectation_type)
    den = np.maximum(np.sqrt(p[..., :, np.newaxis] * p[..., np.newaxis, :]), EPS)
    if csm is None:
        csm = expectation_csm_faithful(coef, expectation_type)
    return _take_nonneg(np.clip(np.abs(csm.imag / den), 0, 1), -3)


def _plv_complex(coef, expectation_type):
    """connectivity.py:897-903: E[s/|s|] (0/0 -> NaN)."""
    with np.errstate(invalid="ignore", divide="ignore"):
        out = expectation_csm_faithful(coef, expectation_type, fcn=lambda s: s / np.abs(s))
    return _take_nonneg(out, -3)


def phase_locking_value(coef, expectation_type="trials_tapers"):
    """connectivity.py:905-931."""
    return np.abs(_plv_complex(coef, expectation_type))


def phase_lag_index(coef, expectation_type="trials_tapers"):
    """connectivity.py:933-980: E[sign Im s], Im forced to 0 on the diagonal."""
    out = expectation_csm_faithful(coef, expectation_type,
                                   fcn=lambda s: np.sign(_zero_diag_imag(s)))
    return _take_nonneg(out.real, -3)


def weighted_phase_lag_index(coef, expectation_type="trials_tapers"):
    """connectivity.py:982-1028: E[Im s] / E[|Im s|], weights < eps -> 1."""
    w = expectation_csm_faithful(coef, expectation_type,
                                 fcn=lambda s: np.abs(_zero_diag_imag(s)))
    w[w < EPS] = 1
    num = expectation_csm_faithful(coef, expectation_type, fcn=_zero_diag_imag)
    return _take_nonneg(num / w, -3)


def debiased_squared_phase_lag_index(coef, expectation_type="trials_tapers"):
    """connectivity.py:1030-1058: (n PLI^2 - 1) / (n - 1)."""
    n = n_observations(coef, expectation_type)
    with np.errstate(invalid="ignore", divide="ignore"):
        return (n * phase_lag_index(coef, expectation_type) ** 2 - 1.0) / (n - 1.0)


def debiased_squared_weighted_phase_lag_index(coef, expectation_type="trials_tapers"):
    """connectivity.py:1060-1127."""
    n = n_observations(coef, expectation_type)
    s_im = expectation_csm_faithful(coef, expectation_type, fcn=_zero_diag_imag) * n
    s_sq = expectation_csm_faithful(coef, expectation_type,
                                    fcn=lambda s: _zero_diag_imag(s) ** 2) * n
    s_abs = expectation_csm_faithful(coef, expectation_type,
                                     fcn=lambda s: np.abs(_zero_diag_imag(s))) * n
    w = s_abs ** 2 - s_sq
    w[w == 0] = np.nan
    return _take_nonneg((s_im ** 2 - s_sq) / w, -3)


def pairwise_phase_consistency(coef, expectation_type="trials_tapers"):
    """connectivity.py:1129-1159."""
    n = n_observations(coef, expectation_type)
    plv_sum = _plv_complex(coef, expectation_type) * n
    with np.errstate(invalid="ignore", divide="ignore"):
        return ((plv_sum * plv_sum.conj() - n) / (n ** 2 - n)).real


# -------------------------------------------------------------- Wilson / spectral Granger
def _ct(x):
    return x.swapaxes(-1, -2).conj()


def minimum_phase_decomposition(csm, tolerance=1e-8, max_iterations=60, return_iterations=False):
    """Wilson spectral factorisation, minimum_phase_decomposition.py:227-322.

    csm: (W, N, c, c) two-sided.  G0 = chol(Re ifft_n(S)[lag 0])^H broadcast over n
    (:48-77; stored in a REAL array, :294-295).  When that Cholesky fails for ANY window of the batch, every window
    starts from the lower Cholesky factor of the mean of 1000 products Z Z^T of standard-normal matrices drawn
    with numpy's GLOBAL generator (:78-93: np.random.seed pins it; no transpose on this branch).
    Iterate A = G^-1 (G^-1 S)^H + I (:184-224); a = ifft_n(A); a[0] *= 1/2; strict lower
    triangle of a[0] = 0; a[n >= (N+1)//2] = 0; A+ = fft_n(a) (:96-142); G <- G A+;
    windows already converged keep G (:310-312); converged(w) = max|G - G_old| < tol
    over all bins/entries of window w (:145-181).
    """
    n_win, n_fft, c = csm.shape[0], csm.shape[-3], csm.shape[-1]
    eye = np.eye(c)
    converged = np.zeros(n_win, dtype=bool)
    try:
        g0 = np.linalg.cholesky(ifft(csm, axis=-3)[..., 0:1, :, :].real).swapaxes(-1, -2)
    except np.linalg.LinAlgError:
        shape = list(csm.shape)
        shape[-3] = 1000
        z = np.random.standard_normal(size=shape)
        g0 = np.linalg.cholesky(np.matmul(z, _ct(z)).mean(axis=-3, keepdims=True))
    G = np.zeros(csm.shape)                 # real on purpose (reference quirk)
    G[...] = g0
    lower = np.tril_indices(c, k=-1)
    n_iter = 0
    for n_iter in range(1, max_iterations + 1):
        old = G.copy()
        X = np.linalg.solve(G, csm)
        A = np.linalg.solve(G, _ct(X)) + eye
        a = ifft(A, axis=-3)
        a[..., 0, :, :] *= 0.5
        a[..., 0, lower[0], lower[1]] = 0
        a[..., (n_fft + 1) // 2:, :, :] = 0
        G = np.matmul(G, fft(a, axis=-3))
        G[converged, ...] = old[converged, ...]
        err = np.abs((G - old).reshape(n_win, -1)).max(axis=1)
        converged = err < tolerance
        if converged.all():
            break
    return (G, n_iter, converged) if return_iterations else G


def _transfer_function(G):
    """connectivity.py:1712-1748: G @ (H0 + lam I)^-1, H0 = Re ifft_n(G)[0], lam = 1e-12 mean(H0^2)."""
    H0 = ifft(G, axis=-3).real[..., 0:1, :, :]
    lam = 1e-12 * np.mean(H0 * H0)
    eye = np.eye(H0.shape[-1], dtype=H0.dtype)
    return np.matmul(G, np.linalg.solve(H0 + lam * eye, eye))


def _noise_covariance(G):
    """connectivity.py:1679-1709: H0 H0^T (real)."""
    H0 = ifft(G, axis=-3).real[..., 0, :, :]
    return np.matmul(H0, H0.swapaxes(-1, -2))


def _mvar_fourier_coefficients(H):
    """connectivity.py:581-589: (H + lam I)^-1 with lam = 1e-12 * mean(|H|^2) over the whole array."""
    lam = 1e-12 * np.mean((np.conj(H) * H).real)
    eye = np.eye(H.shape[-1], dtype=H.dtype)
    return np.linalg.solve(H + lam * eye, eye)


def mvar_quantities(coef, expectation_type="trials_tapers", csm=None):
    """Full C x C Wilson factor of the two-sided CSM and what the MVAR measures derive from it
    (connectivity.py:567-589): G (all N bins), H and A on the non-negative bins, noise covariance."""
    if csm is None:
        csm = expectation_csm_gemm(coef, expectation_type)
    G = minimum_phase_decomposition(csm)
    H = _take_nonneg(_transfer_function(G), -3)
    return dict(G=G, H=H, A=_mvar_fourier_coefficients(H), noise_covariance=_noise_covariance(G))


def _noise_variance(noise_covariance):
    """connectivity.py:1904-1922: diag(Sigma) shaped (..., 1, C, 1): indexed by the ROW channel."""
    return np.diagonal(noise_covariance, axis1=-1, axis2=-2)[..., np.newaxis, :, np.newaxis]


def directed_transfer_function(coef, expectation_type="trials_tapers", q=None):
    """connectivity.py:1237-1270: |H_ij|^2 / sum_j |H_ij|^2."""
    q = q or mvar_quantities(coef, expectation_type)
    p = np.abs(q["H"]) ** 2
    return np.abs(q["H"] / np.sqrt(p.sum(axis=-1, keepdims=True))) ** 2


def directed_coherence(coef, expectation_type="trials_tapers", q=None):
    """connectivity.py:1272-1309: sqrt(nv_i) |H_ij|^2 / sqrt(sum_j nv_i |H_ij|^2), nv = diag(Sigma)."""
    q = q or mvar_quantities(coef, expectation_type)
    nv = _noise_variance(q["noise_covariance"])
    p = np.abs(q["H"]) ** 2
    return np.sqrt(nv) * p / np.sqrt((nv * p).sum(axis=-1, keepdims=True))


def partial_directed_coherence(coef, expectation_type="trials_tapers", q=None):
    """connectivity.py:1311-1364: |A_ij|^2 / sum_i |A_ij|^2, A = MVAR Fourier coefficients."""
    q = q or mvar_quantities(coef, expectation_type)
    p = np.abs(q["A"]) ** 2
    return np.abs(q["A"] / np.sqrt(p.sum(axis=-2, keepdims=True))) ** 2


def generalized_partial_directed_coherence(coef, expectation_type="trials_tapers", q=None):
    """connectivity.py:1366-1400: |A_ij / sqrt(nv_i) / sqrt(sum_i |A_ij|^2 / nv_i)|^2."""
    q = q or mvar_quantities(coef, expectation_type)
    nv = _noise_variance(q["noise_covariance"])
    p = np.abs(q["A"]) ** 2
    return np.abs(q["A"] / np.sqrt(nv) / np.sqrt((p / nv).sum(axis=-2, keepdims=True))) ** 2


def direct_directed_transfer_function(coef, expectation_type="trials_tapers", q=None):
    """connectivity.py:1402-1426: |H_ij| / sqrt(sum_{f,j} |H_ij|^2) * sqrt(PDC_ij)."""
    q = q or mvar_quantities(coef, expectation_type)
    p = np.abs(q["H"]) ** 2
    full = q["H"] / np.sqrt(p.sum(axis=(-1, -3), keepdims=True))
    return np.abs(full) * np.sqrt(partial_directed_coherence(coef, expectation_type, q))


def pairwise_spectral_granger_prediction(coef, expectation_type="trials_tapers", pairs=None):
    """connectivity.py:1161-1191 + :2282-2340 (one 2x2 Wilson problem per channel pair).

    Output [..., i, j] = influence j -> i; diagonal NaN; values <= 0 -> NaN
    (connectivity.py:1751-1779, :1825-1848).
    """
    csm = expectation_csm_gemm(coef, expectation_type)
    p_all = power_two_sided(coef, expectation_type)
    n_sig = csm.shape[-1]
    nn = np.arange(csm.shape[-3] // 2 + 1)
    total_power = np.take(p_all, nn, axis=-2)
    shape = list(csm.shape)
    shape[-3] = nn.size
    out = np.full(shape, np.nan)
    if pairs is None:
        pairs = combinations(range(n_sig), 2)
    for pair in pairs:
        idx = np.array(pair)[:, np.newaxis]
        try:
            G = minimum_phase_decomposition(csm[..., idx, idx.T])
        except np.linalg.LinAlgError:
            out[..., idx, idx.T] = np.nan
            continue
        H = _transfer_function(G)[..., nn, :, :]
        sigma = _noise_covariance(G)
        var = np.diagonal(sigma, axis1=-1, axis2=-2)[..., np.newaxis]
        rot = var.swapaxes(-1, -2) - sigma ** 2 / var
        tp = total_power[..., idx[:, 0]]
        intrinsic = tp[..., np.newaxis] - rot[..., np.newaxis, :, :] * np.abs(H) ** 2
        intrinsic[intrinsic == 0] = EPS
        with np.errstate(invalid="ignore", divide="ignore"):
            gp = np.log(tp[..., np.newaxis]) - np.log(intrinsic)
        gp[gp <= 0] = np.nan
        out[..., idx, idx.T] = gp
    d = np.arange(n_sig)
    out[..., d, d] = np.nan
    return out


# ------------------------------------------------------------------ canonical coherence
def canonical_coherence(coef, group_labels):
    """connectivity.py:745-820, :1953-2032.

    Per group: A (c_g x n_obs) per (w, f) -> U V^H from the thin SVD; per group pair:
    largest singular value of (U V^H)_g (U V^H)_h^H, squared.  (W, F, G, G), NaN diagonal.
    """
    group_labels = np.asarray(group_labels)
    labels = np.unique(group_labels)
    W, _, _, N, _ = coef.shape
    F = N // 2 + 1
    normed = []
    for lab in labels:
        sub = coef[..., :F, :][..., np.isin(group_labels, lab)]       # (W,R,K,F,c)
        A = np.moveaxis(sub.reshape(W, -1, F, sub.shape[-1]), 1, -1)  # (W,F,c,n_obs)
        U, _, Vh = np.linalg.svd(A, full_matrices=False)
        normed.append(np.matmul(U, Vh))
    G = len(labels)
    out = np.full((W, F, G, G), np.nan)
    for a, b in combinations(range(G), 2):
        cross = np.matmul(normed[a], _ct(normed[b]))
        s = np.linalg.svd(cross, compute_uv=False)[..., 0]
        out[..., a, b] = out[..., b, a] = np.abs(s) ** 2
    return out, labels


def global_coherence(coef, max_rank=1):
    """connectivity.py:822-895, :2245-2279: per (window, two-sided bin) the thin SVD of the
    (n_signals, n_trials * n_tapers) coefficient matrix; squared singular values / n_estimates and the
    left singular vectors.  The reference takes scipy's svds when max_rank < n_signals - 1, which
    returns the max_rank largest values in ASCENDING order; otherwise numpy's svd (descending).
    Singular vectors are defined up to a unit phase."""
    W, R, K, N, C = coef.shape
    vals = np.zeros((W, N, max_rank))
    vecs = np.zeros((W, N, C, max_rank), dtype=np.complex128)
    for w in range(W):
        for n in range(N):
            X = coef[w, :, :, n, :].reshape(R * K, C).T
            U, s, _ = np.linalg.svd(X, full_matrices=False)
            g = s[:max_rank] ** 2 / (R * K)
            u = U[:, :max_rank]
            if max_rank < C - 1:
                g, u = g[::-1], u[:, ::-1]
            vals[w, n], vecs[w, n] = g, u
    return vals, vecs


MVAR_MEASURES = {
    "directed_transfer_function": directed_transfer_function,
    "directed_coherence": directed_coherence,
    "partial_directed_coherence": partial_directed_coherence,
    "generalized_partial_directed_coherence": generalized_partial_directed_coherence,
    "direct_directed_transfer_function": direct_directed_transfer_function,
}

MEASURES = {
    "power": power,
    "coherency": coherency,
    "coherence_magnitude": coherence_magnitude,
    "coherence_phase": coherence_phase,
    "imaginary_coherence": imaginary_coherence,
    "phase_locking_value": phase_locking_value,
    "phase_lag_index": phase_lag_index,
    "weighted_phase_lag_index": weighted_phase_lag_index,
    "debiased_squared_phase_lag_index": debiased_squared_phase_lag_index,
    "debiased_squared_weighted_phase_lag_index": debiased_squared_weighted_phase_lag_index,
    "pairwise_phase_consistency": pairwise_phase_consistency,
}
