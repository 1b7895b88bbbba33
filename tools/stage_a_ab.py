"""A/B of stage A variants inside one process (same box, same clocks): SC_MTFFT_DEBUG values given on the command line,
cfg3 volume (128 channels, 1000 trials, 7 tapers, 256-sample half-overlapping windows), alternating, median of 15."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine      # noqa: E402

variants = sys.argv[1:] or ["0", "8"]
dev = torch.device("cuda:0")
SHAPES = ((1024, 256, 1000), (1024, 128, 1000), (2048, 1024, 500))
if os.environ.get("SC_AB_LONG"):
    SHAPES = ((2048, 2048, 500), (4096, 4096, 250))
for (T, L, R) in SHAPES:
    step, K, C = max(1, L // 2), 7, 128
    x = torch.randn((T, R, C), device=dev)
    tap = torch.randn((K, L), device=dev)
    W = (T - L) // step + 1
    gb = (L // 2 + 1) * W * R * K * C * 8 / 1e9
    times = {v: [] for v in variants}
    for rep in range(17):
        for v in variants:
            _lib.set_debug_env("SC_MTFFT_DEBUG", v)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = engine.multitaper_spectra(x, tap, L, step, L, W, "constant")
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out = None
            if rep >= 2:
                times[v].append(dt)
    print(f"N={L}: " + "   ".join(f"dbg={v}: {np.median(times[v]) * 1e3:.3f} ms ({gb / np.median(times[v]) / 1e3:.2f} TB/s)" for v in variants))
    del x
_lib.set_debug_env("SC_MTFFT_DEBUG", None)
