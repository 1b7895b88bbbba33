"""Frequency-band post-processing of the coherency: phase slope index, group delay, delay.

Host-side NumPy on a (..., n_frequencies, n_signals, n_signals) coherency that the device epilogue
produced (reference connectivity.py:1428-1650 and helpers :1653-1676, :2035-2237).  A few kilobytes of
data per call; nothing here belongs on the GPU.
"""
from itertools import combinations

import numpy as np

from .statistics import (adjust_for_multiple_comparisons, coherence_fisher_z_transform,
                         get_normal_distribution_p_values)


def bandpass(data, frequencies, band, axis=-3):
    """Keep the bins strictly inside (band[0], band[1]); everything when band is None (connectivity.py:2035-2073)."""
    if band is None:
        return data, frequencies
    keep = np.flatnonzero((frequencies > band[0]) & (frequencies < band[1]))
    return np.take(data, keep, axis=axis), frequencies[keep]


def independent_frequency_step(frequency_difference, frequency_resolution):
    """Bins between statistically independent estimates (connectivity.py:2076-2100)."""
    if frequency_resolution is None:
        return 1
    return int(np.ceil(frequency_resolution / frequency_difference))


def upper_pairs(n_signals):
    return np.asarray(list(combinations(range(n_signals), 2)), dtype=int).reshape(-1, 2)


def _longest_true_run(mask):
    """Boolean mask of the longest run of consecutive True values (the first one on ties)."""
    out = np.zeros(mask.shape, dtype=bool)
    edges = np.flatnonzero(np.diff(np.concatenate(([0], mask.astype(np.int8), [0]))))
    starts, stops = edges[0::2], edges[1::2]
    if starts.size:
        k = int(np.argmax(stops - starts))
        out[starts[k]:stops[k]] = True
    return out


def _independent_significant(mask, step, min_group_size):
    """connectivity.py:2103-2182: longest significant run, thinned to every `step`-th bin, dropped when fewer than
    `min_group_size` bins remain."""
    run = np.flatnonzero(_longest_true_run(mask))[::step]
    out = np.zeros(mask.shape, dtype=bool)
    if run.size >= min_group_size:
        out[run] = True
    return out


def significant_frequencies(pair_coherency, n_observations, frequency_step=1, significance_threshold=0.05,
                            min_group_size=3, multiple_comparisons_method="Benjamini_Hochberg_procedure"):
    """Bins (axis -2) at which the coherence of each pair (axis -1) is significantly above zero
    (connectivity.py:2185-2237): Fisher z, normal upper-tail p, multiple-comparison control over the whole
    array, then the longest independent run per pair."""
    z = coherence_fisher_z_transform(pair_coherency, n_observations)
    flags = adjust_for_multiple_comparisons(get_normal_distribution_p_values(z), alpha=significance_threshold,
                                            method=multiple_comparisons_method)
    return np.apply_along_axis(_independent_significant, -2, flags, frequency_step, min_group_size)


def phase_slope_index(coherency, frequencies, frequencies_of_interest=None, frequency_resolution=None):
    """Imaginary part of sum over ALL bin pairs a < b in the band of conj(c_a) c_b (connectivity.py:1592-1650,
    :1653-1676).  With the prefix sums P_b = sum_{a<b} c_a this is sum_b conj(P_b) c_b: O(F) instead of O(F^2)."""
    band, band_freq = bandpass(coherency, frequencies, frequencies_of_interest)
    step = independent_frequency_step(frequencies[1] - frequencies[0], frequency_resolution)
    band = band[..., ::step, :, :] if band_freq.shape[0] else band
    prefix = np.cumsum(band, axis=-3) - band
    return np.sum(np.conj(prefix) * band, axis=-3).imag


def _pair_phase(coherency, frequencies, n_observations, frequencies_of_interest, frequency_resolution,
                significance_threshold):
    band, band_freq = bandpass(coherency, frequencies, frequencies_of_interest)
    pairs = upper_pairs(band.shape[-1])
    band = band[..., pairs[:, 0], pairs[:, 1]]
    step = independent_frequency_step(frequencies[1] - frequencies[0], frequency_resolution)
    significant = significant_frequencies(band, n_observations, step, significance_threshold)
    phase = np.unwrap(np.angle(band), axis=-2)
    return phase, significant, band_freq, pairs


def group_delay(coherency, frequencies, n_observations, frequencies_of_interest=None, frequency_resolution=None,
                significance_threshold=0.05):
    """Slope of the unwrapped coherence phase against frequency over the significant bins of the band,
    per channel pair (connectivity.py:1428-1518).  Returns (delay = slope / 2 pi, slope, r_value), each
    (..., n_signals, n_signals), antisymmetric delay/slope, NaN where no significant run exists."""
    phase, significant, band_freq, pairs = _pair_phase(coherency, frequencies, n_observations, frequencies_of_interest,
                                                       frequency_resolution, significance_threshold)
    w = significant.astype(float)
    n = w.sum(axis=-2)
    with np.errstate(invalid="ignore", divide="ignore"):
        fx = band_freq[:, None]
        mx = (w * fx).sum(axis=-2) / n
        my = (w * phase).sum(axis=-2) / n
        dx = (fx - mx[..., None, :]) * w
        dy = (phase - my[..., None, :]) * w
        sxx, syy, sxy = (dx * dx).sum(axis=-2), (dy * dy).sum(axis=-2), (dx * dy).sum(axis=-2)
        slope_p = np.where(n >= 2, sxy / sxx, np.nan)
        r_p = np.where(n >= 2, sxy / np.sqrt(sxx * syy), np.nan)
    n_signals = coherency.shape[-1]
    shape = phase.shape[:-2] + (n_signals, n_signals)
    slope = np.full(shape, np.nan)
    slope[..., pairs[:, 0], pairs[:, 1]] = slope_p
    slope[..., pairs[:, 1], pairs[:, 0]] = -slope_p
    r_value = np.ones(shape)
    r_value[..., pairs[:, 0], pairs[:, 1]] = r_p
    r_value[..., pairs[:, 1], pairs[:, 0]] = r_p
    return slope / (2 * np.pi), slope, r_value


def delay(coherency, frequencies, n_observations, frequencies_of_interest=None, frequency_resolution=None,
          significance_threshold=0.05, n_range=3):
    """Candidate delays (phase + 2 pi k) / 2 pi, k = -n_range .. n_range, per band frequency and channel pair
    (connectivity.py:1520-1590); shape (..., n_frequencies, 2 n_range + 1, n_signals, n_signals), NaN at the
    frequencies where the pair's coherence is not significant.

    The reference computes on a masked array and stores it into a plain one, which keeps the RAW data of the masked
    entries -- the untouched first operand 2 pi k; with its always-NaN one-sample z-score (see
    statistics.coherence_fisher_z_transform) every entry is masked, so its output is the constant 2 pi k
    everywhere.  With options.one_sample_fisher_z == "reference" (default) the non-significant entries carry that same
    raw value, so the output equals the reference's; "unbiased" marks them NaN."""
    from . import options
    phase, significant, _, pairs = _pair_phase(coherency, frequencies, n_observations, frequencies_of_interest,
                                               frequency_resolution, significance_threshold)
    turns = np.arange(-n_range, n_range + 1)
    cand = (2 * np.pi * turns + phase[..., np.newaxis]) / (2 * np.pi)                         # (..., F, P, R)
    masked = 2 * np.pi * turns if options.one_sample_fisher_z == "reference" else np.nan
    cand = np.moveaxis(np.where(significant[..., np.newaxis], cand, masked), -1, -2)          # (..., F, R, P)
    n_signals = coherency.shape[-1]
    out = np.full(cand.shape[:-1] + (n_signals, n_signals), np.nan)
    out[..., pairs[:, 0], pairs[:, 1]] = cand
    out[..., pairs[:, 1], pairs[:, 0]] = -cand
    return out
