#!/usr/bin/env python
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric).

Metric: channel-pair x frequency-bins per second for CSM + coherence (+ wPLI) on the
configuration the metric is quoted on (BASELINE.json configs[2]): 128 channels x 1000 trials
x 1024 samples, NW=4 (7 tapers), sliding 256-sample windows with 128-sample step (W=7,
F=129), coherence_magnitude + weighted_phase_lag_index, expectation over trials x tapers.

One "step" = one full pass of the hot path over the synthetic batch, inputs resident in HBM:
  stage A  window + detrend + taper + FFT + transposed store, one fused HIP kernel (mtfft_long_kernel: two half-workgroups in anti-phase; small problems: mtfft16_kernel)
  stage B  cross-spectral accumulation AND the per-observation |Im s| plane on the 16-bit matrix pipe, one pass
           (round 4: stage A stores every coefficient as two f16 pieces, stage B multiplies three cross terms: sc_fused2.hip)
  (N>1)    reduce-scatter of the accumulator records over RCCL
  stage C  coherence + wPLI epilogue on the owned bins, (N>1) gather of the measures on rank 0
N GPUs: the 1000 trials are sharded over the ranks (strong scaling), one process per GPU.  `python bench.py --gpus N`
starts its own ranks (torch.distributed.run, 127.0.0.1); under the driver's torchrun form it joins the given world.

`--config cfg2 | cfg4 | cfg5` (N = 1) times the other BASELINE.json configurations with their own metric beside the
headline: cfg2 CSM + coherency (pair*bins/s), cfg4 pairwise spectral Granger of all 2016 pairs through the batched
Wilson kernels (channel-pairs/s), cfg5 canonical coherence of 16 groups (bin*group-pairs/s) -- each line carries its
`roofline` and `cpu_baseline`.

Prints ONE JSON line on rank 0 (see the driver contract), with `roofline` for the dominant
kernel (durations from HIP events on the launch stream) and `cpu_baseline` (the NumPy
oracle's faithful per-observation path, single core, on a bounded sample).
"""
import argparse
from ctypes import byref, c_double
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from spectral_connectivity_amd import _lib, engine, parallel  # noqa: E402
from spectral_connectivity_amd.transforms import _make_tapers  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32 dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (the pipe the one-pass stage B actually runs on)
F64_PEAK_TFLOPS = 78.6        # fp64 vector = fp64 matrix rate
# HBM bytes per launch from the PMC counters: a PMC pass cannot run inside this process, so `roofline.traffic` is read
# from the committed summary of the rocprofv3 --pmc passes over THIS command (tools/profile_round.py writes it next to
# the kernel-trace stats; it records the source hash of the kernels it measured) and is null when that file is
# missing or was measured on other kernel sources.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r06_hbm_traffic.json")


def kernel_source_hash():
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "spectral_connectivity_amd", "csrc")
    for name in ("sc_fused.hip", "sc_fused2.hip", "sc_fused_common.h", "sc_mtfft.hip", "sc_mtfft_long.hip", "sc_mtfft_bfly.h", "sc_measure.hip", "sc_stage.h", "sc_common.h",
                 "sc_wilson_pair.hip", "sc_wilson_fft.h"):
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def measured_traffic(config, stage):
    try:
        rec = json.load(open(TRAFFIC_FILE))
    except (OSError, ValueError):
        return None, None
    if rec.get("kernel_source_hash") != kernel_source_hash():
        return None, "profiles/r06_hbm_traffic.json was measured on other kernel sources"
    v = rec.get(config, {}).get(stage)
    return (float(v) if v is not None else None), rec.get("source")

CONFIGS = {
    # name: T, R, C, NW, L, step; kind = what a step computes after stages A and B
    "cfg3": dict(T=1024, R=1000, C=128, NW=4.0, L=256, step=128, tone=60.0, kind="measures",
                 label="128ch x 1000 trials x 1024 samples, NW=4 (7 tapers), 256-pt windows step 128 "
                       "(W=7, F=129), coherence_magnitude + weighted_phase_lag_index"),
    "cfg2": dict(T=1024, R=100, C=32, NW=3.0, L=1024, step=1024, tone=40.0, kind="coherency",
                 label="32ch x 100 trials x 1024 samples, NW=3 (5 tapers), single window, CSM + coherency"),
    "cfg4": dict(T=4096, R=200, C=64, NW=3.0, L=4096, step=4096, tone=30.0, kind="granger",
                 label="64ch x 200 trials x 4096 samples, NW=3 (5 tapers), single window, "
                       "pairwise_spectral_granger_prediction of all 2016 channel pairs (batched 2x2 Wilson, fp64)"),
    "cfg5": dict(T=1024, R=500, C=256, NW=3.0, L=1024, step=1024, tone=30.0, kind="canonical",
                 label="256ch x 500 trials x 1024 samples, NW=3 (5 tapers), single window, canonical_coherence "
                       "between 16 groups of 16 channels (513 bins x 120 group pairs, fp64)"),
}
FS = 1000.0


def synth(cfg, r_lo, r_hi, device, seed):
    """SURVEY 8(d) synthetic input: white noise + shared sinusoid, per-channel phase 2*pi*c/C."""
    T, C = cfg["T"], cfg["C"]
    g = torch.Generator(device=device)
    g.manual_seed(seed * 1000003 + r_lo)
    x = torch.randn((T, r_hi - r_lo, C), generator=g, device=device, dtype=torch.float32)
    t = torch.arange(T, device=device, dtype=torch.float32) / FS
    ph = 2 * np.pi * torch.arange(C, device=device, dtype=torch.float32) / C
    x += 0.5 * torch.sin(2 * np.pi * cfg["tone"] * t[:, None, None] + ph[None, None, :])
    return x


def one_step(x, h, cfg, geom, planes, world, exchange=None):
    """One pass of the hot path.  Stage durations come from the library's own hipEvent timers (sc_last_timing: every
    entry point brackets its launches on the stream it was given); `exchange` collects the collective breakout of the
    N > 1 path."""
    L, step, N, W = geom
    sp = engine.multitaper_spectra(x, h, L, step, N, W, "constant", planes_hint=planes)
    if world > 1 or os.environ.get("SC_BENCH_FORCE_SHARDED") == "1":     # (the switch: the N > 1 code path on one rank,
        # with SC_FORCE_EXCHANGE=1 through the collectives too -- what a rank's step costs beside the transfers)
        # trial shards: accumulate -> reduce-scatter -> epilogue -> gather on rank 0, pipelined over frequency
        # groups so that only the last group's exchange is exposed (parallel.sharded_measures)
        coh, wpli = parallel.sharded_measures(sp, planes, [_lib.M_COHERENCE_MAGNITUDE, _lib.M_WPLI],
                                              n_groups=int(os.environ.get("SC_BENCH_GROUPS", "4")),
                                              equal_shards=True, timing=exchange)   # R % world == 0: asserted in main()
        return coh, wpli
    # fold=False: the split-bin partial records come back as ONE [n_parts, n_bins, floats_per_bin] tensor and the epilogue sums them
    # while it reads -- the path Connectivity takes too (connectivity.py: _accumulators), checked against the folded form and the
    # float64 reference at this size in tests/test_gpu_full_depth.py::test_cfg3_bench_chain_full_depth
    accum, n_obs = engine.accumulate(sp, "trials_tapers", planes, fold=False)
    del sp
    coh, wpli = engine.measure_multi(accum, cfg["C"], planes, n_obs, [_lib.M_COHERENCE_MAGNITUDE, _lib.M_WPLI])
    return coh, wpli


def chain_check(x, h, cfg, geom, planes):
    """Outside the timed region: the timed chain (partial records + parts-summing epilogue) against the folded form of the same
    launch, bit for bit, and a sanity bound on the values."""
    L, step, N, W = geom
    sp = engine.multitaper_spectra(x, h, L, step, N, W, "constant", planes_hint=planes)
    parts, n_obs = engine.accumulate(sp, "trials_tapers", planes, fold=False)
    folded, _ = engine.accumulate(sp, "trials_tapers", planes)
    which = [_lib.M_COHERENCE_MAGNITUDE, _lib.M_WPLI]
    a = engine.measure_multi(parts, cfg["C"], planes, n_obs, which)
    b = engine.measure_multi(folded, cfg["C"], planes, n_obs, which)
    same = all(bool(torch.equal(u.nan_to_num(), v.nan_to_num())) for u, v in zip(a, b))
    off = ~torch.eye(cfg["C"], dtype=torch.bool, device=x.device)
    sane = bool(((a[0][:, off] >= 0) & (a[0][:, off] <= 1)).all()) and bool((a[1].abs() <= 1 + 1e-6).all())
    return {"spectra_format": "planes (f16 pieces)" if sp.P is not None else "complex64",
            "n_partial_records": int(parts.shape[0]) if parts.dim() == 3 else 1,
            "parts_epilogue_equals_folded_record_bitwise": same, "values_in_range": sane}


def api_pass(x, cfg, geom):
    """The same workload through the PUBLIC classes, series resident in HBM: Multitaper(device tensor) ->
    Connectivity.from_multitaper(dtype=complex64) -> coherence_magnitude() -> weighted_phase_lag_index() -- the BASELINE order, the
    one that froze complex64 spectra before round 5.  Device time from the library's own timers (every kernel of the pass), wall
    time including the two downloads (the API returns NumPy arrays)."""
    import spectral_connectivity_amd as sc
    L, step, N, W = geom
    kw = dict(sampling_frequency=FS, time_halfbandwidth_product=cfg["NW"], n_time_samples_per_window=L, n_time_samples_per_step=step)
    dev_ms, wall_ms, fmt = [], [], None
    for _ in range(4):
        torch.cuda.synchronize()
        _lib.last_timing()
        t0 = time.perf_counter()
        c = sc.Connectivity.from_multitaper(sc.Multitaper(x, **kw), dtype=np.complex64)
        coh = c.coherence_magnitude()
        wpli = c.weighted_phase_lag_index()
        wall_ms.append((time.perf_counter() - t0) * 1e3)
        stages = {}
        for name, ms in _lib.last_timing():
            stages[name] = stages.get(name, 0.0) + ms
        total = sum(stages.values())
        ck = c_double(0.0)
        _lib.load().sc_debug_fused2_clock(byref(ck))
        stages["stage_b_clock_ghz"] = ck.value                 # (the clock the part sustained inside stage B of THIS pass)
        dev_ms.append((total, stages))
        fmt = "planes (f16 pieces)" if c._spectra.P is not None else "complex64"
        del c, coh, wpli
    every = [{k: round(v, 3) for k, v in st.items()} for _, st in dev_ms]
    dev_ms.sort(key=lambda t: t[0])
    wall_ms.sort()
    total, stages = dev_ms[1]
    return {"device_ms": round(total, 4), "wall_ms": round(wall_ms[1], 3), "spectra_format": fmt, "every_pass": every,
            "stage_ms": {k: round(v, 4) for k, v in stages.items()},
            "is": "Multitaper(series in HBM) -> Connectivity.from_multitaper(dtype=complex64) -> coherence_magnitude() -> "
                  "weighted_phase_lag_index(); device_ms = the library's hipEvent timers over every kernel of the pass, wall_ms with the "
                  "host-side parameter logic, taper generation and the two downloads of float64 / float32 results; second fastest of 4"}


def cpu_baseline_strong(cfg, geom, budget_trials=16):
    """A CPU path that is NOT the reference's algorithm but the best NumPy restructuring of it: one-sided spectra,
    the cross-spectral matrix as one batched GEMM per (window, bin) on all BLAS threads, and the |Im s| plane by
    blocked vectorised arithmetic -- so that the GPU / CPU ratio is not inflated by the reference's per-observation
    outer products.  Same float64 results (checked against the faithful path in tests/test_oracle_golden.py)."""
    from oracle import spectral_oracle as so
    L, step, N, W = geom
    x = synth(cfg, 0, budget_trials, "cpu", seed=3).numpy().astype(np.float64)
    t0 = time.perf_counter()
    coef, _ = so.multitaper_fft(x, fs=FS, NW=cfg["NW"], n_time_samples_per_window=L, n_time_samples_per_step=step)
    F = N // 2 + 1
    X = coef[:, :, :, :F, :]                                         # (W, R, K, F, C) non-negative bins only
    Wn, R, K, _, C = X.shape
    Xo = np.ascontiguousarray(np.moveaxis(X, 3, 1)).reshape(Wn, F, R * K, C)     # (W, F, obs, C)
    n = R * K
    S = np.matmul(np.swapaxes(Xo, -1, -2), Xo.conj()) / n            # E[x_i conj x_j], batched zgemm
    P = np.real(np.einsum("wfii->wfi", S))
    coh = np.abs(S) ** 2 / np.maximum(P[..., :, None] * P[..., None, :], 1e-300)
    re, im = Xo.real, Xo.imag
    wsum = np.zeros((Wn, F, C, C))
    for o0 in range(0, n, 16):                                       # |Im(x_i conj x_j)| summed over observations
        a, b2 = im[:, :, o0:o0 + 16, :, None], re[:, :, o0:o0 + 16, None, :]
        c2, d2 = re[:, :, o0:o0 + 16, :, None], im[:, :, o0:o0 + 16, None, :]
        wsum += np.abs(a * b2 - c2 * d2).sum(axis=2)
    wsum /= n
    wsum[wsum < 2.220446049250313e-16] = 1
    wpli = S.imag / wsum
    dt = time.perf_counter() - t0
    return dt, float(coh[0, 1, 0, 1]), float(wpli[0, 1, 0, 1])


def cpu_baseline(cfg, geom, budget_trials=4):
    """Time the oracle's faithful (reference op-for-op) path on `budget_trials` trials."""
    from oracle import spectral_oracle as so
    L, step, N, W = geom
    x = synth(cfg, 0, budget_trials, "cpu", seed=3).numpy().astype(np.float64)
    t0 = time.perf_counter()
    for r in range(budget_trials):          # one trial at a time: bounds the (W,1,K,N,C,C) temporary to 3.3 GB
        coef, _ = so.multitaper_fft(x[:, r:r + 1], fs=FS, NW=cfg["NW"], n_time_samples_per_window=L,
                                    n_time_samples_per_step=step)
        so.coherence_magnitude(coef)
        so.weighted_phase_lag_index(coef)
    dt = time.perf_counter() - t0
    return dt


def run_side_config(args, cfg, device):
    """cfg2 / cfg4 / cfg5 of BASELINE.json on one GPU: the same stages A and B, then the configuration's own consumer --
    coherency, the batched 2x2 Wilson factorisation + Granger prediction of every channel pair, or canonical coherence
    between the channel groups.  One JSON line with the configuration's own metric, `roofline` for the kernel that takes
    the largest share of the step, and `cpu_baseline` (the oracle on a bounded sample)."""
    from oracle import spectral_oracle as so
    T, R, C, L, step = cfg["T"], cfg["R"], cfg["C"], cfg["L"], cfg["step"]
    N, W = L, int(np.floor(T / step - L / step + 1))
    F = N // 2 + 1
    tapers = _make_tapers(L, FS, cfg["NW"], int(np.floor(2 * cfg["NW"] - 1)))
    K = tapers.shape[1]
    h = torch.from_numpy(np.ascontiguousarray(tapers.T / FS, dtype=np.float32)).to(device)
    x = synth(cfg, 0, R, device, seed=3)
    kind = cfg["kind"]
    planes = _lib.PLANE_CSM
    n_obs_loc = R * K
    pairs = np.array([(i, j) for i in range(C) for j in range(i + 1, C)], dtype=np.int32)
    groups = [np.arange(g * 16, (g + 1) * 16) for g in range(C // 16)]
    info = {}

    def one():
        sp = engine.multitaper_spectra(x, h, L, step, N, W, "constant")
        accum, n_obs = engine.accumulate(sp, "trials_tapers", planes)
        del sp
        if kind == "coherency":
            return engine.measure(accum, C, planes, n_obs, _lib.M_COHERENCY)
        if kind == "granger":
            out, n_it, _, summary = engine.granger_pairwise(accum, W, F, N, C, planes, n_obs, pairs)
            info["wilson"], info["n_iter"] = summary, n_it
            return out
        out, n_fail = engine.canonical_coherence(accum, C, planes, n_obs, groups)
        info["canonical_failures"] = n_fail
        return out

    for _ in range(args.warmup):
        one()
    torch.cuda.synchronize()
    _lib.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timing = _lib.last_timing()
    _lib.timing_enable(False)
    ms_per_step = elapsed / args.steps * 1e3
    stage_ms = {}
    for name, ms in timing:
        stage_ms[name] = stage_ms.get(name, 0.0) + ms / args.steps
    # A pass whose kernels are each a few tens of microseconds is bound by the host's launches (cfg2: 82 us of kernels in a
    # 113 us step): engine.GraphedMeasures -- the library's captured form of the pass (stage A, stage B, epilogue in ONE
    # hipGraph, replayed per step on the object's own buffers) -- checked against the eager pass before it is timed.  (Stage
    # times above come from the eager passes: events cannot sit inside a replayed graph.)
    launch = {"mode": "eager"}
    if kind == "coherency" and ms_per_step < 1.0 and os.environ.get("SC_BENCH_GRAPH", "1") == "1":
        try:
            ref = one()
            g = engine.GraphedMeasures((T, R, C), h, L, step, N, "constant", "trials_tapers", [_lib.M_COHERENCY])
            out_g, = g(x)
            torch.cuda.synchronize()
            same = bool(torch.equal(torch.view_as_real(out_g).nan_to_num(), torch.view_as_real(ref).nan_to_num()))
            if same:
                for _ in range(args.warmup):
                    g()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    g()
                torch.cuda.synchronize()
                g_ms = (time.perf_counter() - t0) / args.steps * 1e3
                launch = {"mode": "engine.GraphedMeasures: hipGraph replay of the pass (stage A, stage B, epilogue captured once by the library); "
                                  "bit-identical to the eager pass",
                          "eager_ms_per_step": round(ms_per_step, 5), "graph_ms_per_step": round(g_ms, 5)}
                if g_ms < ms_per_step:
                    ms_per_step, elapsed = g_ms, g_ms * args.steps * 1e-3
            else:
                launch = {"mode": "eager", "graph": "replay differed from the eager pass: not used"}
        except Exception as exc:                     # capture refused (an allocation or a synchronisation inside the pass): stay eager
            launch = {"mode": "eager", "graph": "capture failed: " + str(exc).splitlines()[0][:160]}
            torch.cuda.synchronize()

    tri_flops = 8.0 * n_obs_loc * (C * (C + 1) / 2) * W * F
    iters = info.get("wilson", (0, 0, 0))[0]
    if "n_iter" in info:
        info["wilson_iter_sum"] = int(info["n_iter"].sum().item())
    n_gp = len(groups) * (len(groups) - 1) // 2
    # algorithmic work per launch
    stage_model = {
        "mtfft_fused": ("hbm", 4.0 * T * R * C + 8.0 * F * W * R * K * C),
        "fused_stage_b": ("mfma", tri_flops),
        "csm_mfma": ("mfma", tri_flops),
        "measure_epilogue": ("hbm", (2 * 4.0 * C * (C + 1) / 2 + 8.0 * C * C) * W * F),
        # pairwise Granger, resident form (sc_wilson_pair.hip): everything between the records and the converged factor happens in
        # registers / LDS, so the bound is the fp64 arithmetic -- per (problem, iteration) four 4096-point complex transforms
        # (5 N log2 N flop each; the zero / unused halves are pruned: counted in full here) + ~255 flop per non-negative bin for
        # A = G^-1 S G^-H + I, the conjugate-symmetry split, G <- G A+ and the convergence norm; iterations = what the problems
        # really ran (sum over the problems of their own counts)
        "granger_pairwise": ("f64", (4 * 5.0 * N * np.log2(N) + 255.0 * (N // 2 + 1)) * max(info.get("wilson_iter_sum", 0), 1)),
        # canonical coherence (approximate fp64 flop model, per (bin, group pair) of 16-channel groups: two 16^3 complex
        # whitening products + M M^H (3 x 32.8 kflop) + ~6 cyclic Jacobi sweeps of 120 rotations on 16 x 16 (~0.74 Mflop))
        "canonical_coherence": ("f64", 0.84e6 * W * F * n_gp),
    }

    def stage_roof(name):
        b, wk = stage_model[name]
        d = stage_ms[name] * 1e-3
        peak, unit, scale = {"mfma": (MFMA_F32_PEAK_TFLOPS, "TFLOP/s", 1e12), "f64": (F64_PEAK_TFLOPS, "TFLOP/s", 1e12),
                             "hbm": (HBM_PEAK_GBS, "GB/s", 1e9)}[b]
        return {"bound": "mfma" if b == "f64" else b, "achieved": round(wk / d / scale, 3), "peak": peak, "unit": unit,
                "frac": round(wk / d / scale / peak, 4), "kernel_ms": round(stage_ms[name], 4)}

    dominant = max((k for k in stage_model if k in stage_ms), key=lambda k: stage_ms[k])
    roofline = dict(stage_roof(dominant), entry_point=dominant, traffic=None,
                    stage_ms={k: round(v, 4) for k, v in stage_ms.items()},
                    stages={k: stage_roof(k) for k in stage_ms if k in stage_model})
    if dominant == "granger_pairwise":
        resident = os.environ.get("SC_GRANGER_KERNEL") != "batched"
        roofline["kernel"] = "wilson_pair_kernel (+ pair_lag0 / pair_consts / pair_granger)" if resident else "k_update + causal_fft_pair_kernel + k_flags per iteration"
        roofline["note"] = (f"2 x 2 Wilson iteration of {len(pairs) * W} (window, pair) problems x {N // 2 + 1} non-negative bins, {info.get('wilson_iter_sum', 0)} "
                            f"problem-iterations (most for one problem: {iters}); fp64 vector peak (the matrix and the vector pipe share the units); flop "
                            "model in bench.py; over the whole entry point (lag-0 covariances, the factorisation, the prediction)")
        t_bytes, t_src = measured_traffic(args.config, "granger_pairwise")
        roofline["traffic"], roofline["traffic_source"] = t_bytes, t_src
        # what has to move: the pairs' cross-spectra out of the records once (4 floats per pair and bin) and the prediction out once;
        # the resident kernel adds the factor of the non-negative bins (written once, read once by the prediction kernel)
        roofline["algorithmic_bytes"] = {"spectra_two_sided_2x2_f64": 16.0 * 4 * N * len(pairs) * W,
                                         "records_read": 4.0 * 4 * (N // 2 + 1) * len(pairs) * W, "prediction_written": 8.0 * W * (N // 2 + 1) * C * C}
    elif dominant == "canonical_coherence":
        roofline["note"] = "approximate fp64 flop model of the Cholesky whitening + parallel Jacobi per (bin, group pair); fp64 vector peak"

    if kind == "coherency":
        units, unit = float(W * F * C * C), "channel-pair*freq-bins/s"
        metric = "channel-pair*freq-bins/s for CSM+coherency"
    elif kind == "granger":
        units, unit = float(len(pairs) * W), "channel-pairs/s"
        metric = f"channel-pairs/s for pairwise spectral Granger prediction ({N}-sample windows, {F} bins per pair)"
    else:
        units, unit = float(W * F * n_gp), "bin*group-pairs/s"
        metric = "frequency-bin*group-pairs/s for canonical coherence (16 groups of 16 channels)"
    value = units / (elapsed / args.steps)

    cpu = None
    if not args.no_cpu_baseline:
        # the oracle on a bounded sample of the same workload, one core; what the sample was is said in `sample`
        if kind == "coherency":
            xs = synth(cfg, 0, 8, "cpu", seed=3).numpy().astype(np.float64)
            t1 = time.perf_counter()
            coef, _ = so.multitaper_fft(xs, fs=FS, NW=cfg["NW"], n_time_samples_per_window=L)
            so.coherency(coef)
            dt = time.perf_counter() - t1
            cpu_value, sample = units / (dt * R / 8), (f"oracle faithful path (per-observation outer product + mean) on 8 of "
                                                        f"{R} trials, linear in trials")
        elif kind == "granger":
            xs = synth(dict(cfg, C=4), 0, 20, "cpu", seed=3).numpy().astype(np.float64)
            t1 = time.perf_counter()
            coef, _ = so.multitaper_fft(xs, fs=FS, NW=cfg["NW"], n_time_samples_per_window=L)
            so.pairwise_spectral_granger_prediction(coef)
            dt = time.perf_counter() - t1
            cpu_value, sample = 6.0 / dt, ("oracle: 6 pairs (4 channels) x 20 trials of the same window length -- the cost of a "
                                           "pair is its Wilson iteration over the 4096 bins, independent of the trial count")
        else:
            Ts = 128
            xs = synth(dict(cfg, T=Ts), 0, R, "cpu", seed=3).numpy().astype(np.float64)
            t1 = time.perf_counter()
            coef, _ = so.multitaper_fft(xs, fs=FS, NW=cfg["NW"], n_time_samples_per_window=Ts)
            so.canonical_coherence(coef, np.arange(C) // 16)
            dt = time.perf_counter() - t1
            cpu_value, sample = (Ts // 2 + 1) * n_gp / dt, (f"oracle (per-bin SVDs) with all {R} trials and {C} channels on {Ts}-sample "
                                                            f"windows ({Ts // 2 + 1} of {F} bins): the cost is linear in the bins")
        cpu = {"value": round(cpu_value, 3), "unit": unit, "cores": 1, "kind": "port",
               "measured_seconds_on_sample": round(dt, 3), "sample": sample + f"; os.cpu_count()={os.cpu_count()}"}

    return ({
        "metric": metric, "value": value, "unit": unit, "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32" if kind == "coherency" else "f32 spectra + records, f64 " + ("Wilson iteration" if kind == "granger" else "whitening + Jacobi"),
        "data": "synthetic",
        "config": {"workload": cfg["label"], "name": args.config, "trials_total": R, "n_tapers": K, "n_windows": W,
                   "n_freq_bins": F, "units_per_step": units,
                   **({"wilson_iterations": iters, "wilson_not_converged": info["wilson"][1]} if kind == "granger" else {})},
        "roofline": roofline, "launch": launch, "cpu_baseline": cpu})


def side_configs(args, device):
    """The other GPU configurations of BASELINE.json (configs[1], [3], [4]) behind the headline's timed region, each with its own
    metric, stage times, roofline and CPU baseline -- `side_configs` of the default line, so that the driver's one run carries
    every configuration (python bench.py --config cfgN prints any of them as a line of its own)."""
    import copy
    out = {}
    for name in ("cfg2", "cfg4", "cfg5"):
        a = copy.copy(args)
        a.config, a.steps, a.warmup = name, 10, 3
        try:
            t0 = time.perf_counter()
            line = run_side_config(a, dict(CONFIGS[name]), device)
            line["wall_seconds_in_bench"] = round(time.perf_counter() - t0, 2)
            for k in ("n_gpus", "higher_is_better", "vs_baseline", "data", "scaling"):
                line.pop(k, None)
            out[name] = line
        except Exception as exc:                      # a side configuration never costs the headline its line
            out[name] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    return out


def self_launch(n_ranks):
    """Re-run this command under torch.distributed.run with n_ranks local ranks (rendezvous on 127.0.0.1).  With fewer
    visible GPUs than ranks (a debug run on a 1-GPU box) the ranks share devices over gloo -- said on stderr and in the
    line's "backend" field: such a number exercises the N > 1 control flow, it is not a scaling measurement."""
    import socket
    import subprocess
    env = dict(os.environ)
    n_dev = torch.cuda.device_count()
    if n_dev < n_ranks and "SC_BENCH_BACKEND" not in env:
        print(f"bench.py: {n_ranks} ranks requested, {n_dev} GPU(s) visible: ranks share devices over gloo "
              "(control-flow rehearsal, not a scaling measurement)", file=sys.stderr)
        env["SC_BENCH_BACKEND"] = "gloo"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--trials", type=int, default=None, help="override total trial count (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f64", action="store_true", help="skip the float64-engine side measurement")
    ap.add_argument("--timed-only", action="store_true",
                    help="nothing but the warm-up and the timed steps (no CPU baselines, float64 engine, end-to-end / API / chain-check "
                         "passes): the command the rocprofv3 summaries under profiles/ are taken from, so that their per-kernel "
                         "averages are averages over the steps the line reports")
    args = ap.parse_args()
    if args.timed_only:
        args.no_cpu_baseline = args.no_f64 = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU), the same
        # command the driver's torchrun form runs; rank 0 prints the one JSON line, the exit code is the job's.
        sys.exit(self_launch(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # SC_BENCH_BACKEND=gloo (debug only): exercise the N>1 control flow on a box with fewer GPUs than ranks
    backend = os.environ.get("SC_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    _lib.require_gpu()

    cfg = dict(CONFIGS[args.config])
    if args.trials:
        cfg["R"] = args.trials
    if cfg["kind"] != "measures":
        assert world == 1, f"--config {args.config} is a single-GPU line; the scaling bench is cfg3"
        print(json.dumps(run_side_config(args, cfg, device)))
        return
    assert cfg["R"] % world == 0, "trials must divide evenly over ranks"
    r_lo, r_hi = parallel.shard_bounds(cfg["R"], world, rank)
    T, C, L, step = cfg["T"], cfg["C"], cfg["L"], cfg["step"]
    N = L
    W = int(np.floor(T / step - L / step + 1))
    F = N // 2 + 1
    geom = (L, step, N, W)
    K_req = int(np.floor(2 * cfg["NW"] - 1))
    tapers = _make_tapers(L, FS, cfg["NW"], K_req)                    # (L, K) float64, host, once
    K = tapers.shape[1]
    h = torch.from_numpy(np.ascontiguousarray(tapers.T / FS, dtype=np.float32)).to(device)
    x = synth(cfg, r_lo, r_hi, device, seed=3)
    planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step(x, h, cfg, geom, planes, world)
    sync()
    _lib.timing_enable(True)                  # hipEvent pairs inside the library from here on (sc_timing.hip)
    exchange = [] if world > 1 else None
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]     # per-step times for the median (no sync inside)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        one_step(x, h, cfg, geom, planes, world, exchange)
        marks[i + 1].record()
    sync()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    clock = c_double(0.0)
    _lib.load().sc_debug_fused2_clock(byref(clock))          # in-kernel cycle / real-time counters of the last stage-B launch
    timing = _lib.last_timing()
    _lib.timing_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    units = W * F * C * C                                   # channel-pair x frequency bins
    value = units / (elapsed / args.steps)

    # per-stage durations: the library's hipEvent pairs (sc_last_timing), summed per entry point, averaged over steps
    stage_ms = {}
    for name, ms in timing:
        stage_ms[name] = stage_ms.get(name, 0.0) + ms / args.steps
    R_loc = r_hi - r_lo
    n_obs_loc = R_loc * K
    # algorithmic work per launch on this rank (DESIGN.md "Roofline")
    stage_model = {
        "mtfft_fused": ("hbm", 4.0 * T * R_loc * C + 8.0 * F * W * R_loc * K * C),
        "planes_scales": ("hbm", 4.0 * T * R_loc * C),          # the scan of the series for the f16 scales
        "taper_windows": ("hbm", 4.0 * T * R_loc * C + 4.0 * N * W * R_loc * K * C),
        "rocfft_r2c": ("hbm", 4.0 * N * W * R_loc * K * C + 8.0 * F * W * R_loc * K * C),
        "fused_stage_b": ("mfma", 8.0 * n_obs_loc * (C * (C + 1) / 2) * W * F),
        "csm_mfma": ("mfma", 8.0 * n_obs_loc * (C * (C + 1) / 2) * W * F),
        "nonlinear_valu": ("hbm", 8.0 * F * W * R_loc * K * C + 4.0 * W * F * C * (C + 1) / 2),
        # (one launch for both measures: the three record planes are read once)
        "measure_epilogue": ("hbm", (3 * 4.0 * C * (C + 1) / 2 + 2 * 4.0 * C * C) * W * F / world),
    }
    dominant = max((k for k in stage_model if k in stage_ms), key=lambda k: stage_ms[k])
    bound, work = stage_model[dominant]
    dur_s = stage_ms[dominant] * 1e-3
    if bound == "mfma":
        achieved, peak, unit = work / dur_s / 1e12, MFMA_F32_PEAK_TFLOPS, "TFLOP/s"
    else:
        achieved, peak, unit = work / dur_s / 1e9, HBM_PEAK_GBS, "GB/s"
    def stage_roof(name):
        b, wk = stage_model[name]
        d = stage_ms[name] * 1e-3
        if b == "mfma":
            return {"bound": b, "achieved": round(wk / d / 1e12, 3), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(wk / d / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)}
        return {"bound": b, "achieved": round(wk / d / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(wk / d / 1e9 / HBM_PEAK_GBS, 4)}

    traffic, traffic_src = measured_traffic(args.config, dominant) if world == 1 else (None, None)
    KERNEL_OF = {"fused_stage_b": "fused2_kernel (its partial records are summed by the epilogue; fused_combine_kernel only on the fold=True path)", "mtfft_fused": "mtfft_long_kernel (anti-phase half-workgroups; mtfft16_kernel below 256 items)",
                 "measure_epilogue": "measure_tile_kernel"}
    roofline = {"kernel": KERNEL_OF.get(dominant, dominant), "entry_point": dominant, "bound": bound,
                "achieved": round(achieved, 3), "peak": peak,
                "unit": unit, "frac": round(achieved / peak, 4),
                # HBM bytes per launch from the rocprofv3 PMC passes over this command (FETCH_SIZE x2 gfx950 correction +
                # WRITE_SIZE), read from profiles/r03_hbm_traffic.json; null when absent / measured on other sources
                "traffic": traffic, "traffic_source": traffic_src,
                "kernel_ms": round(stage_ms[dominant], 4),
                "frac_is": ("f32-equivalent flops of the Hermitian rank-n_obs update, upper triangle only "
                            "(8*n_obs*C(C+1)/2 per bin), over the f32 MFMA peak" if bound == "mfma" else
                            "algorithmic bytes over the HBM peak"),
                "note": ("the kernel runs the update as three f16 cross terms (two-piece split written by stage A) on the 16-bit "
                         "matrix pipe and also produces the per-observation |Im s| plane in the same launch; the other "
                         "normalisations are beside `frac`"),
                # the clock the part sustained inside the dominant kernel (s_memtime / s_memrealtime in three workgroups), and
                # `frac` against the peak AT THAT CLOCK (the 157.3 TF peak assumes 2.4 GHz)
                **({} if not (bound == "mfma" and clock.value > 0) else {
                    "sustained_clock_ghz": round(clock.value, 3),
                    "frac_at_clock": round(achieved / (peak * clock.value / 2.4), 4)}),
                **({} if bound != "mfma" else {
                    "flops_triangle": work,
                    # SURVEY 8(d) counts the full C x C matrix (the mirror is free on this design): same time, twice the flops
                    "flops_full_matrix": 8.0 * n_obs_loc * C * C * W * F,
                    "frac_full_matrix": round(8.0 * n_obs_loc * C * C * W * F / dur_s / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                    # the pipe the products really run on: 3 f16 cross terms per f32-equivalent product (6 bf16 ones before round 4)
                    "frac_f16_pipe": round(3.0 * work / dur_s / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                    # against the pipe the kernel really uses: every matrix instruction it issues, at that instruction's own rate
                    "pipe": (lambda nt, nb32, obs_bins: (lambda csm_f, abs_f: {
                        "unit": "TFLOP/s", "peak": MFMA_BF16_PEAK_TFLOPS, "achieved": round((csm_f + abs_f) / dur_s / 1e12, 2),
                        "frac": round((csm_f + abs_f) / dur_s / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                        "pipe_busy_frac": round((csm_f / (MFMA_BF16_PEAK_TFLOPS * 1e12) + abs_f / (0.5 * MFMA_BF16_PEAK_TFLOPS * 1e12)) / dur_s, 4),
                        "counts": "issued f16 matrix flops: cross-spectra 12 x v_mfma_f32_16x16x32_f16 per upper 16x16 tile and 32 observations "
                                  "(three cross terms x {rr, ii, ir, (-r)i}, diagonal tiles in full) + one v_mfma_f32_32x32x8_f16 per "
                                  "observation and upper 32x32 block for the per-observation Im s (K = 8: the four piece products of "
                                  "Im Re and of (-Re) Im; a half-rate instruction: pipe_busy_frac prices it at 1.25 PF)"})(
                        obs_bins * nt * 12 * 16384.0 / 32.0, obs_bins * (nb32 * (nb32 + 1) // 2) * 16384.0))(
                        ((C + 15) // 16) * ((C + 15) // 16 + 1) // 2, (C + 31) // 32, float(n_obs_loc) * W * F)}),
                # the whole step against both rooflines of SURVEY section 8(d) (the binding one is the larger time):
                # algorithmic bytes of the two-pass design over the HBM peak, triangle-only CSM flops over the f32 MFMA peak
                "whole_path": (lambda t_hbm, t_mfma: {
                    "t_hbm_ms": round(t_hbm, 4), "t_mfma_ms": round(t_mfma, 4),
                    "hbm_frac": round(t_hbm / ms_per_step, 4), "mfma_frac": round(t_mfma / ms_per_step, 4)})(
                    (4.0 * T * R_loc * C + 2 * 8.0 * F * W * R_loc * K * C + (8.0 + 2 * 4.0) * W * F * C * C / world)
                    / (HBM_PEAK_GBS * 1e9) * 1e3,
                    8.0 * n_obs_loc * (C * (C + 1) / 2) * W * F / (MFMA_F32_PEAK_TFLOPS * 1e12) * 1e3),
                "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
                "stages": {k: stage_roof(k) for k in stage_ms if k in stage_model}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        n_sample = 4
        dt = cpu_baseline(cfg, geom, n_sample)
        cpu = {"value": round(units / (dt * cfg["R"] / n_sample), 3), "unit": "channel-pair*freq-bins/s",
               "cores": 1, "kind": "port",
               "measured_seconds_on_sample": round(dt, 3),
               "sample": (f"oracle faithful path (per-observation outer product + mean, float64 NumPy, "
                          f"single thread) on {n_sample} of {cfg['R']} trials of the same workload; cost is "
                          f"exactly linear in trials, value = units / (t_sample * {cfg['R']}/{n_sample}); "
                          f"os.cpu_count()={os.cpu_count()}")}

    cpu_strong = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        n_sample = 16
        dt, _, _ = cpu_baseline_strong(cfg, geom, n_sample)
        try:
            from threadpoolctl import threadpool_info
            threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
        except Exception:
            threads = os.cpu_count()
        cpu_strong = {"value": round(units / (dt * cfg["R"] / n_sample), 3), "unit": "channel-pair*freq-bins/s",
                      "cores": int(threads), "kind": "port", "measured_seconds_on_sample": round(dt, 3),
                      "sample": (f"restructured NumPy path (one-sided spectra, batched-GEMM cross-spectral matrix on all BLAS "
                                 f"threads, blocked vectorised |Im s| plane; float64) on {n_sample} of {cfg['R']} trials, linear "
                                 f"in trials; the |Im s| plane is single-threaded NumPy arithmetic, BLAS threads = {threads}")}

    check = api = None
    if rank == 0 and world == 1 and not args.timed_only:
        check = chain_check(x, h, cfg, geom, planes)
        _lib.timing_enable(True)
        api = api_pass(x, cfg, geom)
        _lib.timing_enable(False)

    # end to end, NumPy in -> NumPy out (page-locked host buffers: upload of the float32 series, the step,
    # download of both measures), a few passes after the timed region; never `value`
    e2e_ms = None
    if rank == 0 and world == 1 and not args.timed_only:
        x_host = torch.empty(x.shape, dtype=torch.float32, pin_memory=True)
        x_host.copy_(x)
        torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            t1 = time.perf_counter()
            xd = x_host.to(device, non_blocking=True)
            coh_d, wpli_d = one_step(xd, h, cfg, geom, planes, world)
            coh_h, wpli_h = engine.to_host(coh_d), engine.to_host(wpli_d)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t1) * 1e3)
            del xd, coh_d, wpli_d, coh_h, wpli_h
        e2e_ms = round(sorted(ts[1:])[len(ts[1:]) // 2], 3)
        del x_host

    # Beside the line (never `value`): the float64 engine -- the reference's default dtype, the path that meets 1e-5 on
    # every element at this depth -- on the same workload, a few passes after the timed region.
    f64 = None
    if rank == 0 and world == 1 and not args.no_f64:
        del x, h
        torch.cuda.empty_cache()
        xd = synth(cfg, r_lo, r_hi, device, seed=3).to(torch.float64)
        hd = torch.from_numpy(np.ascontiguousarray(tapers.T / FS)).to(device)

        def f64_step():
            sp = engine.multitaper_spectra_f64(xd, hd, L, step, N, W, "constant")
            accum, n_obs = engine.accumulate(sp, "trials_tapers", planes)
            del sp
            return engine.measure_multi(accum, C, planes, n_obs, [_lib.M_COHERENCE_MAGNITUDE, _lib.M_WPLI], wide=True)
        for _ in range(2):
            f64_step()
        torch.cuda.synchronize()
        _lib.timing_enable(True)
        n64 = 5
        t1 = time.perf_counter()
        for _ in range(n64):
            f64_step()
        torch.cuda.synchronize()
        dt64 = (time.perf_counter() - t1) / n64
        st64 = {}
        for name, ms in _lib.last_timing():
            st64[name] = st64.get(name, 0.0) + ms / n64
        _lib.timing_enable(False)
        # stage B of this engine against the fp64 units (matrix rate = vector rate = 78.6 TFLOP/s, one set of units): the
        # instruction slots its arithmetic occupies, an fma slot counted as 2 flop like the peak counts it
        nb16, nb64 = (C + 15) // 16, (C + 63) // 64
        n_bins_obs = float(W * F) * float(n_obs_loc)
        csm_products = 4            # Re: ar ar + ai ai, Im: ai ar - ar ai  (a three-product form was tried: profiles/r04_f64_three_products.txt)
        slots_csm = n_bins_obs * (nb16 * (nb16 + 1) // 2) * 256 * csm_products * 2
        # (a diagonal 64 x 64 block computes the 36 of its 64 sub-tiles that touch the upper triangle: sc_f64.hip)
        slots_plane = n_bins_obs * (nb64 * (nb64 - 1) // 2 * 4096 + nb64 * 2304) * 3 * 2
        t_b64 = st64.get("accumulate_f64", 0.0) * 1e-3
        roof64 = None
        if t_b64 > 0:
            ach = (slots_csm + slots_plane) / t_b64 / 1e12
            roof64 = {"kernel": "csm_f64_kernel + nonlinear_f64_block_kernel (accumulate_f64)", "bound": "mfma", "achieved": round(ach, 2),
                      "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / F64_PEAK_TFLOPS, 4), "kernel_ms": round(t_b64 * 1e3, 4),
                      "frac_is": "fp64 instruction slots of stage B (%d real matrix products per 16x16 tile and observation; mul + fma + "
                                 "|.|-add per pair and observation of the 64x64 blocks), 2 flop a slot, over the fp64 peak the matrix "
                                 "and the vector pipe share" % csm_products}
        f64 = {"ms_per_step": round(dt64 * 1e3, 3), "value": round(units / dt64, 1), "dtype": "f64", "steps": n64,
               "stage_ms": {k: round(v, 4) for k, v in st64.items()}, "roofline": roof64,
               "note": "Connectivity(dtype=complex128): float64 transform, fp64 matrix-core CSM, fp64 VALU |Im s| plane, "
                       "float64 measures; reported beside the float32 line, not as the metric"}

    side = None
    if rank == 0 and world == 1 and not args.timed_only and os.environ.get("SC_BENCH_SIDE_CONFIGS", "1") == "1":
        try:
            del xd, hd
        except NameError:
            pass
        torch.cuda.empty_cache()
        side = side_configs(args, device)
    if rank == 0:
        print(json.dumps({
            "metric": "channel-pair*freq-bins/s for CSM+coherence(+wPLI)",
            "value": value, "unit": "channel-pair*freq-bins/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # the median step (hipEvents between the steps of the same timed loop), the observation-normalised
            # rate W F C^2 R K / t (config-independent work rate) and the end-to-end time with both PCIe legs
            "ms_per_step_median": round(median_ms, 4), "value_at_median": units / (median_ms * 1e-3),
            "pair_bin_obs_per_s": units * cfg["R"] * K / (elapsed / args.steps),
            "e2e_ms": e2e_ms, "e2e_is": "NumPy (pinned) float32 series in -> coherence + wPLI as NumPy out, median of 3",
            "api_ms": None if api is None else api["device_ms"], "api": api, "chain_check": check,
            "config": {"workload": cfg["label"], "name": args.config, "trials_total": cfg["R"],
                       "trials_per_gpu": R_loc, "n_tapers": K, "n_windows": W, "n_freq_bins": F,
                       "units_per_step": units, "parallelism": f"trials sharded over {world} GPU(s)"},
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_restructured": cpu_strong,
            "float64_engine": f64,
            # BASELINE.json configs[1], [3], [4] on this GPU, behind the timed region of the headline (side_configs())
            "side_configs": side,
            "backend": None if world == 1 else ("rccl" if backend == "nccl" else backend + " (ranks share GPUs: rehearsal only)"),
            # N > 1: time inside the RCCL collectives of one step on the exchange stream (reduce-scatter of the records,
            # gather of the measures) and the part of the exchange + epilogue the launch stream had to wait for
            "exchange": None if exchange is None else {
                "collective_ms": round(sum(e["collective_ms"] for e in exchange) / max(len(exchange), 1), 4),
                "exposed_exchange_ms": round(sum(e["exposed_ms"] for e in exchange) / max(len(exchange), 1), 4),
                "bytes_reduced_per_rank": exchange[-1]["bytes_reduced"] if exchange else None,
                "n_frequency_groups": exchange[-1]["n_groups"] if exchange else None,
                "reduce_scatter": parallel.exchange_note(),
                # what the measured numbers should be judged against: bytes per xGMI link and their time at the assumed link rate
                "model": parallel.exchange_model(4.0 * W * F * 3 * 256 * (((C + 15) // 16) * ((C + 15) // 16 + 1) // 2),   # 3 record planes
                                                 2 * 4.0 * W * F * C * C, world)},
        }))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
