"""Minimum-phase (Wilson) spectral factorisation on the GPU.

Drop-in for ``spectral_connectivity.minimum_phase_decomposition.minimum_phase_decomposition``
(reference minimum_phase_decomposition.py:227-322).  1x1 and 2x2 cross-spectral matrices -- the
sizes the hot path (pairwise spectral Granger prediction) uses -- advance together in fp64 through
``sc_wilson_factor_f64`` (batched closed-form 2x2 solves, fused causal transform pair along the frequency axis);
larger systems (up to ``sc_mvar_max_signals()`` = 512 signals) go through ``sc_mvar_factor_f64``
(register-resident Gauss-Jordan solves, one workgroup per window and frequency bin).  There is no CPU fallback.
"""
import ctypes
from ctypes import byref
from logging import getLogger

import numpy as np

logger = getLogger(__name__)


def minimum_phase_decomposition(cross_spectral_matrix, tolerance=1e-8, max_iterations=60):
    """Minimum-phase square root G of a Hermitian spectral density, S = G G^H.

    cross_spectral_matrix : complex array, shape (n_time, ..., n_fft_samples, c, c), two-sided in
        frequency.  Returns an array of the same shape (complex128).
    Differences from the reference: every leading-axis problem stops at its own convergence (the
    reference freezes a window once it has converged -- same iterate); a 1 x 1 / 2 x 2 problem whose lag-0
    covariance is not positive definite starts from the identity -- the expectation of the reference's random
    Wishart start (minimum_phase_decomposition.py:78-93) -- with the same warning.
    """
    import torch

    from . import _lib
    from .engine import _ptr, _stream, check_max_iterations
    max_iterations = check_max_iterations(max_iterations)
    _lib.require_gpu()
    lib = _lib.load()
    csm = np.asarray(cross_spectral_matrix)
    if csm.ndim < 3 or csm.shape[-1] != csm.shape[-2]:
        raise ValueError("cross_spectral_matrix must have shape (..., n_fft_samples, n_signals, n_signals)")
    c, N = csm.shape[-1], csm.shape[-3]
    lead = csm.shape[:-3]
    P = int(np.prod(lead)) if lead else 1
    if c > 2:
        from . import engine
        if c > lib.sc_mvar_max_signals():
            raise NotImplementedError(f"the HIP Wilson kernels factorise up to {lib.sc_mvar_max_signals()} x "
                                      f"{lib.sc_mvar_max_signals()} spectra; got n_signals={c}")
        dev = torch.device("cuda", torch.cuda.current_device())
        out = np.empty((P, N, c, c), dtype=np.complex128)
        flat = np.ascontiguousarray(csm.reshape(P, N, c, c).astype(np.complex128))
        step = 4096
        for p0 in range(0, P, step):
            n = min(step, P - p0)
            G, _, status, (_, not_conv, fallback) = engine.mvar_factor(
                n, N, c, spectra=torch.from_numpy(flat[p0:p0 + n]).to(dev), tolerance=tolerance,
                max_iterations=max_iterations)
            if fallback:
                logger.warning("Computing the initial conditions using the Cholesky failed. "
                               f"Using the identity as initial condition ({fallback} problems).")
            if not_conv:
                logger.warning(f"Maximum iterations reached. {n - not_conv} of {n} converged")
            out[p0:p0 + n] = G.cpu().numpy()
        return out.reshape(csm.shape)
    flat = csm.reshape(P, N, c, c).astype(np.complex128)
    S = np.empty((P, 4, N), dtype=np.float64)
    S[:, 0] = flat[:, :, 0, 0].real
    if c == 2:
        S[:, 1] = flat[:, :, 1, 1].real
        S[:, 2] = flat[:, :, 0, 1].real
        S[:, 3] = flat[:, :, 0, 1].imag
    else:                                   # embed s as diag(s, 1): G = diag(g, 1)
        S[:, 1] = 1.0
        S[:, 2:] = 0.0
    dev = torch.device("cuda", torch.cuda.current_device())
    out = np.empty((P, N, c, c), dtype=np.complex128)
    step = max(1, min(P, (4 << 30) // (N * 160)))
    for p0 in range(0, P, step):
        n = min(step, P - p0)
        S_d = torch.from_numpy(S[p0:p0 + n]).to(dev)
        nbytes = ctypes.c_size_t()
        _lib.check(lib.sc_granger_workspace_bytes(1, n, N, byref(nbytes)), "sc_granger_workspace_bytes")
        work = torch.empty((nbytes.value,), dtype=torch.uint8, device=dev)
        G_d = torch.empty((n, 4, N), dtype=torch.complex128, device=dev)
        n_iter = torch.empty((n,), dtype=torch.int32, device=dev)
        status = torch.empty((n,), dtype=torch.int32, device=dev)
        summary = (ctypes.c_int32 * 3)(0, 0, 0)
        _lib.check(lib.sc_wilson_factor_f64(_ptr(S_d), n, N, tolerance, max_iterations, _ptr(work), nbytes.value,
                                            _ptr(G_d), _ptr(n_iter), _ptr(status), summary, _stream()),
                   "sc_wilson_factor_f64")
        if summary[2]:
            logger.warning("Computing the initial conditions using the Cholesky failed. "
                           f"Using the identity as initial condition ({summary[2]} problems).")
        if summary[1]:
            logger.warning(f"Maximum iterations reached. {n - summary[1]} of {n} converged")
        G = G_d.cpu().numpy()                                   # (n, 4, N)
        if c == 2:
            out[p0:p0 + n] = np.moveaxis(G, 1, -1).reshape(n, N, 2, 2)
        else:
            out[p0:p0 + n, :, 0, 0] = G[:, 0]
    return out.reshape(csm.shape)
