"""The hot path end to end (stage A -> stage B -> epilogue: coherence + wPLI) at the cfg3 data volume for several window lengths,
through the engine's own choice of kernels (planes format where it applies); stage durations from the library's timers.
    python tools/e2e_lengths.py [N ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine      # noqa: E402

dev = torch.device("cuda:0")
PL = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
lengths = [int(v) for v in sys.argv[1:]] or [256, 250, 200, 500, 1000, 1024]
lib = _lib.load()
print("#     N  W     R | step ms (median of 7) | stage durations (ms) | format")
for N in lengths:
    K, C = 7, 128
    step = N // 2
    Wt = max(1, round(1792 / N))
    T = step * (Wt + 1)
    W = (T - N) // step + 1
    R = int(1000 * 1024 / T)
    x = torch.randn((T, R, C), device=dev)
    t = torch.arange(T, device=dev) / 1000.0
    x += 0.5 * torch.sin(2 * np.pi * 60.0 * t[:, None, None] + 2 * np.pi * torch.arange(C, device=dev)[None, None, :] / C)
    tap = torch.randn((K, N), device=dev) / 30

    def one():
        sp = engine.multitaper_spectra(x, tap, N, step, N, W, "constant", planes_hint=PL)
        fmt = "planes" if sp.P is not None else "complex64"
        accum, n_obs = engine.accumulate(sp, "trials_tapers", PL, fold=False)
        del sp
        out = engine.measure_multi(accum, C, PL, n_obs, [_lib.M_COHERENCE_MAGNITUDE, _lib.M_WPLI])
        return fmt, out

    for _ in range(3):
        fmt, out = one(); del out
    torch.cuda.synchronize()
    ts = []
    lib.sc_timing_enable(1)
    _lib.last_timing()
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fmt, out = one()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        del out
    stages = {}
    for name, ms in _lib.last_timing():
        stages.setdefault(name, []).append(ms)
    lib.sc_timing_enable(0)
    st = ", ".join(f"{k} {np.median(v):.3f}" for k, v in stages.items())
    print(f"N={N:5d} {W:2d} {R:5d} | {np.median(ts) * 1e3:6.2f} | {st} | {fmt}", flush=True)
    del x
