"""TEST INFRASTRUCTURE: the oracle's mathematics (oracle/spectral_oracle.py, i.e. reference transforms.py:1147-1171,
:1311-1405 and connectivity.py:447-526, :612-702, :982-1028) restated in float64 torch so that it can run at the FULL
BASELINE sizes, where the NumPy oracle's per-observation temporaries (3.3 TB at configs[2]) cannot exist.  It shares no
code with the product: plain torch.fft / einsum in float64 on the device, chunked over (window, bin).  Every user pins
it against the NumPy oracle at a reduced trial count first (`pin_against_oracle`), so a full-size comparison is a
comparison with the oracle's arithmetic, not with a second opinion."""
import numpy as np
import torch

EPS = float(np.finfo(np.float64).eps)


def spectra_fp64(x, tapers, fs, L, step, n_fft, detrend="constant"):
    """(T, R, C) host array -> one-sided complex128 device tensor X[f, w, r, k, c], like the reference's
    `fft(detrended window * tapers, n=N) / fs` (transforms.py:1377-1405); tapers = (L, K) already scaled by sqrt(fs)."""
    xd = torch.as_tensor(np.asarray(x), device="cuda").to(torch.float64)
    T, R, C = xd.shape
    W = int(np.floor(T / step - L / step + 1))
    h = torch.as_tensor(np.asarray(tapers, dtype=np.float64), device="cuda")        # (L, K)
    K = h.shape[1]
    F = n_fft // 2 + 1
    X = torch.empty((F, W, R, K, C), dtype=torch.complex128, device="cuda")
    for w in range(W):
        seg = xd[w * step:w * step + L]                                              # (L, R, C)
        if detrend == "constant":
            seg = seg - seg.mean(dim=0, keepdim=True)
        elif detrend is not None:
            raise NotImplementedError(detrend)
        for k in range(K):
            y = seg * h[:, k, None, None]
            X[:, w, :, k, :] = torch.fft.rfft(y, n=n_fft, dim=0) / fs
    return X


def sums_fp64(X, want_abs=True, bins_per_chunk=2):
    """X[f, w, r, k, c] -> (sum_o x_i conj x_j [w, f, c, c] complex128, sum_o |Im(x_i conj x_j)| [w, f, c, c] float64
    or None), o running over trials x tapers: the two un-normalised expectations coherence and wPLI need."""
    F, W, R, K, C = X.shape
    n = R * K
    Xo = X.reshape(F, W, n, C)
    csm = torch.empty((W, F, C, C), dtype=torch.complex128, device=X.device)
    ab = torch.empty((W, F, C, C), dtype=torch.float64, device=X.device) if want_abs else None
    for f0 in range(0, F, bins_per_chunk):
        xs = Xo[f0:f0 + bins_per_chunk]
        csm[:, f0:f0 + bins_per_chunk] = torch.einsum("fwoc,fwod->wfcd", xs, xs.conj())
        if want_abs:
            for f in range(f0, min(F, f0 + bins_per_chunk)):
                for w in range(W):
                    re, im = Xo[f, w].real, Xo[f, w].imag                             # (n, C)
                    acc = torch.zeros((C, C), dtype=torch.float64, device=X.device)
                    step = max(1, (1 << 25) // (C * C))
                    for o0 in range(0, n, step):
                        d = im[o0:o0 + step, :, None] * re[o0:o0 + step, None, :] \
                            - re[o0:o0 + step, :, None] * im[o0:o0 + step, None, :]
                        acc += d.abs_().sum(0)
                    ab[w, f] = acc
    return csm, ab


def measures_fp64(csm_sum, abs_sum, n_obs):
    """Un-normalised sums -> dict of float64 NumPy arrays (power, coherency, coherence_magnitude,
    weighted_phase_lag_index) with the reference's eps clamps and diagonals (connectivity.py:612-702, :982-1028)."""
    S = (csm_sum / n_obs).cpu().numpy()
    C = S.shape[-1]
    idx = np.arange(C)
    P = np.real(S[..., idx, idx])
    norm = np.sqrt(P[..., :, None] * P[..., None, :])
    norm[norm < EPS] = EPS
    coherency = S / norm
    coherency[..., idx, idx] = np.nan
    out = dict(power=P, coherency=coherency, coherence_magnitude=np.clip(np.abs(coherency) ** 2, 0, 1))
    if abs_sum is not None:
        im = S.imag.copy()
        im[..., idx, idx] = 0
        wgt = (abs_sum / n_obs).cpu().numpy()
        wgt[..., idx, idx] = 0
        wgt[wgt < EPS] = 1
        out["weighted_phase_lag_index"] = im / wgt
    return out


def relative_error_report(got, ref, floor=1e-3):
    """Elementwise relative error over the entries with |ref| > floor * max|ref| (NaNs must coincide).
    Returns (max, 99.9th percentile, fraction of entries above the floor)."""
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    a, b = got[ok], ref[ok]
    big = np.abs(b) > floor * np.abs(b).max()
    rel = np.abs(a[big] - b[big]) / np.abs(b[big])
    return float(rel.max()), float(np.quantile(rel, 0.999)), float(big.mean())
