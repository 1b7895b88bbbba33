"""Small-sample statistics for coherence and power estimates (host-side NumPy/SciPy).

Mirrors the public functions of the reference's ``statistics`` module (reference statistics.py:21-480)
so that code written against it keeps working.  These run on a few kilobytes of already-reduced
measures (coherency per frequency and channel pair); they are not part of the device hot path.
"""
import numpy as np
import scipy.special
import scipy.stats

np.seterr(invalid="ignore")

_ONE_MINUS_EPS = 1.0 - np.finfo(float).eps


def Benjamini_Hochberg_procedure(p_values, alpha=0.05):
    """False-discovery-rate control: reject every hypothesis whose p-value is at most the largest sorted
    p_(k) lying under the line alpha * k / m (all p-values form one family).  reference statistics.py:21-59."""
    p = np.array(p_values)
    ranked = np.sort(p, axis=None)
    m = ranked.size
    under = np.flatnonzero(ranked <= alpha * np.arange(1, m + 1) / m) if m else np.array([], dtype=int)
    cutoff = ranked[under[-1]] if under.size else -1.0
    return p <= cutoff


def Bonferroni_correction(p_values, alpha=0.05):
    """Family-wise error control: p <= alpha / (number of tests).  reference statistics.py:62-92."""
    p = np.asarray(p_values)
    return p <= alpha / p.size


MULTIPLE_COMPARISONS = {
    "Benjamini_Hochberg_procedure": Benjamini_Hochberg_procedure,
    "Bonferroni_correction": Bonferroni_correction,
}


def adjust_for_multiple_comparisons(p_values, alpha=0.05, method="Benjamini_Hochberg_procedure"):
    """Boolean mask of the tests that stay significant (reference statistics.py:101-144)."""
    return MULTIPLE_COMPARISONS[method](p_values, alpha=alpha)


def coherence_bias(n_observations):
    """Bias of arctanh|coherency| with 2 n degrees of freedom: 1 / (2 n - 2).  reference statistics.py:250-288."""
    return 1.0 / (2 * n_observations - 2)


def coherence_fisher_z_transform(coherency1, n_obs1, coherency2=0, n_obs2=0):
    """z-score of a coherence (or of the difference of two): Fisher transform arctanh|c| minus its bias,
    scaled by the standard deviation sqrt(bias1 + bias2); magnitudes >= 1 are pulled just below 1.
    reference statistics.py:147-203."""
    def fisher(c, bias):
        mag = np.array(np.abs(c), dtype=float)
        mag[mag >= 1] = _ONE_MINUS_EPS
        return np.arctanh(mag) - bias

    # One-sample test (n_obs2 = 0): the reference evaluates coherence_bias(0) = -1/2 for the absent second
    # estimate, which makes sqrt(bias1 + bias2) -- and with it every one-sample z-score, group_delay() and the
    # significance mask of delay() -- NaN.  That is the default here too (drop-in, pinned by
    # tests/golden/f11_post.npz); options.one_sample_fisher_z = "unbiased" gives the absent sample no bias.
    from . import options
    b1 = coherence_bias(n_obs1)
    if n_obs2 or options.one_sample_fisher_z == "reference":
        b2 = coherence_bias(n_obs2)
    else:
        b2 = 0.0
    with np.errstate(invalid="ignore"):
        return (fisher(coherency1, b1) - fisher(coherency2, b2)) / np.sqrt(b1 + b2)


def get_normal_distribution_p_values(data, mean=0, std_deviation=1):
    """Upper-tail probability of a normal variate (reference statistics.py:206-247)."""
    return 1 - scipy.stats.norm.cdf(np.asarray(data), loc=mean, scale=std_deviation)


def coherence_rate_adjustment(firing_rate_condition1, firing_rate_condition2, spike_power_spectrum,
                              homogeneous_poisson_noise=0, dt=1):
    """Factor that corrects spike-field coherence for a firing-rate change between two conditions
    (Aoi et al. 2015).  Multiply the coherence of condition 1 by it.  reference statistics.py:291-351."""
    ratio = firing_rate_condition2 / firing_rate_condition1
    rate_term = ((1 / ratio - 1) * firing_rate_condition1 + homogeneous_poisson_noise / ratio ** 2) * dt ** 2
    return 1 / np.sqrt(1 + rate_term / spike_power_spectrum)


def power_confidence_intervals(n_tapers, power=1, ci=0.95):
    """Chi-square confidence band of a multitaper power estimate with 2 K degrees of freedom
    (Kramer & Eden 2016).  Returns (lower, upper).  reference statistics.py:354-399."""
    dof = 2 * n_tapers
    upper = dof / scipy.stats.chi2.ppf(1 - ci, dof) * power
    lower = dof / scipy.stats.chi2.ppf(ci, dof) * power
    return lower, upper


def power_bias(n_observations):
    """Bias of log power: digamma(2 n) - log(2 n).  reference statistics.py:402-415."""
    dof = 2 * n_observations
    return scipy.special.psi(dof) - np.log(dof)


def power_variance(n_observations):
    """Variance of log power: trigamma(2 n).  reference statistics.py:418-444."""
    return scipy.special.polygamma(1, 2 * n_observations)


def power_fisher_z_transform(spectrum1, n_obs1, spectrum2=0, n_obs2=0):
    """z-score of a log power (or of the difference of two), bias corrected.  reference statistics.py:447-480."""
    z1 = np.log(spectrum1) - power_bias(n_obs1)
    z2 = np.log(spectrum2) - power_bias(n_obs2)
    return (z1 - z2) / np.sqrt(power_variance(n_obs1) + power_variance(n_obs2))
