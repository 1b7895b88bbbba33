"""Synthetic multivariate autoregressive data for examples and tests (host-side NumPy).

Same call as the reference's ``simulate.simulate_MVAR`` (reference simulate.py): with the same
``random_state`` it draws the same innovations and therefore returns the same series.
"""
import numpy as np


def simulate_MVAR(coefficients, noise_covariance=None, n_time_samples=100, n_trials=1, n_burnin_samples=100,
                  random_state=None):
    """x[t] = sum_k A_k x[t - k] + e[t],  e ~ N(0, noise_covariance), for every trial.

    coefficients : (n_lags, n_signals, n_signals), ``coefficients[k - 1][i, j]`` is the weight of signal j at lag k
        on signal i.  Returns (n_time_samples, n_trials, n_signals) after discarding ``n_burnin_samples``.
    """
    coefficients = np.asarray(coefficients, dtype=float)
    n_lags, n_signals, _ = coefficients.shape
    cov = np.eye(n_signals) if noise_covariance is None else noise_covariance
    rng = random_state if isinstance(random_state, np.random.Generator) else np.random.default_rng(random_state)
    n_total = n_time_samples + n_burnin_samples
    x = rng.multivariate_normal(np.zeros(n_signals), cov, size=(n_total, n_trials))      # innovations, in place
    for t in range(n_lags, n_total):
        for k in range(n_lags):
            x[t] += x[t - k - 1] @ coefficients[k].T
    return x[n_burnin_samples:]
