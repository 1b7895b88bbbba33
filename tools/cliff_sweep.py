"""Timing sweep over NEIGHBOURING parameters of the stage-D / section 8(f) measures (round 6): a kernel family usually changes at a
size boundary (16 / 17 channels a group, 64 / 65 signals, a power-of-two window or not) -- a cliff there is a path nobody measured.
ms per call (second call of two), public classes, float64 records where the measure is fp64.  Usage: python tools/cliff_sweep.py [what...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spectral_connectivity_amd as sc      # noqa: E402

what = set(sys.argv[1:]) or {"global", "granger", "mvar", "expectation"}


def timed(fn, reps=2):
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return 1e3 * dt, out


def series(T, R, C, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((T, R, C)).astype(np.float32)
    x[1:] += 0.5 * x[:-1]
    x[:, :, 1:] += 0.3 * x[:, :, :-1]
    return x


if "global" in what:
    for C in (32, 48, 64, 65, 96, 128):
        m = sc.Multitaper(series(1024, 200, C), sampling_frequency=1000.0, time_halfbandwidth_product=3)
        c = sc.Connectivity.from_multitaper(m, dtype=np.complex64)
        c.coherence_magnitude()
        ms, (vals, _) = timed(lambda: c.global_coherence(max_rank=2))
        print(f"global coherence    C={C:4d}, {vals.shape[1]} bins: {ms:8.1f} ms")

if "granger" in what:
    for L in (4096, 4000, 2048, 2000, 1024, 1000, 500):
        m = sc.Multitaper(series(4096, 50, 32, 1), sampling_frequency=1000.0, time_halfbandwidth_product=3,
                          n_time_samples_per_window=L, n_time_samples_per_step=L)
        c = sc.Connectivity.from_multitaper(m)
        c.coherence_magnitude()
        ms, g = timed(lambda: c.pairwise_spectral_granger_prediction())
        it = c._last_wilson["iterations"] if getattr(c, "_last_wilson", None) else -1
        print(f"pairwise Granger    32 ch (496 pairs), window {L:5d} ({g.shape[0]} windows): {ms:8.1f} ms, {it} iterations")

if "mvar" in what:
    for C in (8, 16, 32, 48, 64, 65, 80):
        m = sc.Multitaper(series(1792, 40, C, 2), sampling_frequency=500.0, time_halfbandwidth_product=3, n_time_samples_per_window=256)
        def run():
            c = sc.Connectivity.from_multitaper(m)
            return c.directed_transfer_function(), c
        ms, (d, c) = timed(run)
        print(f"full Wilson + DTF   C={C:4d}, 7 windows x 256 bins: {ms:8.1f} ms, {c._last_wilson['iterations']} iterations")

if "expectation" in what:
    x = series(1024, 200, 64, 3)
    for et in ("trials_tapers", "trials", "tapers", "time_trials_tapers", "time_trials", "time_tapers", "time"):
        m = sc.Multitaper(x, sampling_frequency=1000.0, time_halfbandwidth_product=4, n_time_samples_per_window=256, n_time_samples_per_step=128)
        def run():
            c = sc.Connectivity.from_multitaper(m, expectation_type=et, dtype=np.complex64)
            return c.coherence_magnitude(), c.weighted_phase_lag_index()
        ms, (coh, w) = timed(run)
        print(f"coherence + wPLI    64 ch x 200 trials, expectation over {et:20s}: {ms:8.1f} ms, out {coh.shape}")

if "hot" in what or "hot64" in what:
    # the hot path itself (float32 engine, coherence + wPLI through the public classes, series resident in HBM, the library's own timers):
    # one parameter of BASELINE configs[2] varied at a time; ms of device time per pass and per GB of one-sided complex64 spectra
    from spectral_connectivity_amd import _lib
    base = dict(T=1024, R=250, C=128, L=256, step=128, NW=4, detrend="constant")
    variants = [("base (cfg3 / 4 trials)", {})]
    variants += [(f"C={c}", dict(C=c)) for c in (40, 44, 64, 100, 127, 129, 130, 192, 256, 258, 306, 307, 512)]
    variants += [(f"window={l} step={s}", dict(L=l, step=s)) for l, s in ((256, 256), (256, 64), (128, 64), (64, 32), (32, 16), (100, 50), (300, 150),
                                                                          (384, 192), (768, 384), (1024, 1024))]
    variants += [(f"T={t} window={t}", dict(T=t, L=t, step=t, R=64)) for t in (4096, 8192, 5000, 6000)]
    variants += [(f"R={r}", dict(R=r)) for r in (249, 10, 1)]
    variants += [(f"NW={nw}", dict(NW=nw)) for nw in (1.5, 2, 8)]
    variants += [(f"detrend={d}", dict(detrend=d)) for d in (None, "linear")]
    _lib.timing_enable(True)
    for name, kv in variants:
        p = dict(base, **kv)
        x = torch.from_numpy(series(p["T"], p["R"], p["C"], 5)).cuda()
        try:
            def run():
                m = sc.Multitaper(x, sampling_frequency=1000.0, time_halfbandwidth_product=p["NW"], n_time_samples_per_window=p["L"],
                                  n_time_samples_per_step=p["step"], detrend_type=p["detrend"])
                c = sc.Connectivity.from_multitaper(m, dtype=np.complex128 if "hot64" in what else np.complex64)
                return c.coherence_magnitude(), c.weighted_phase_lag_index(), m
            run()
            torch.cuda.synchronize(); _lib.last_timing()
            coh, w, m = run()
            torch.cuda.synchronize()
            tm = _lib.last_timing()
            dev = sum(v for _, v in tm)
            W = coh.shape[0] if coh.ndim == 4 else 1
            F = coh.shape[-3]
            K = m.tapers.shape[1]
            gb = 8.0 * W * p["R"] * K * F * p["C"] / 1e9
            stages = {}
            for k, v in tm:
                stages[k] = stages.get(k, 0.0) + v
            top = ", ".join(f"{k} {v:.2f}" for k, v in sorted(stages.items(), key=lambda kv: -kv[1])[:4])
            print(f"hot path  {name:28s}: device {dev:7.2f} ms for {gb:6.2f} GB of spectra = {dev / gb:6.2f} ms/GB   [{top}]")
        except Exception as e:      # noqa: BLE001
            print(f"hot path  {name:28s}: {type(e).__name__}: {str(e)[:120]}")

if "mvar_windows" in what:
    for C in (16, 64, 128):
        for L in (256, 250, 1024, 1000):
            m = sc.Multitaper(series(7 * L, 40, C, 2), sampling_frequency=500.0, time_halfbandwidth_product=3, n_time_samples_per_window=L)
            def run():
                c = sc.Connectivity.from_multitaper(m)
                return c.directed_transfer_function(), c
            ms, (d, c) = timed(run)
            print(f"full Wilson + DTF   C={C:4d}, 7 windows x {L:5d} bins: {ms:8.1f} ms, {c._last_wilson['iterations']} iterations")
