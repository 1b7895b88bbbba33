"""What the prologue of sc_mtfft_mixed.hip costs by part: SC_MTFFT_DEBUG=3 (prologue + the slots' barriers only) with detrend None /
constant / linear, at the cfg3 volume."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine      # noqa: E402

dev = torch.device("cuda:0")


def timed(f, reps=5):
    out = f(); out = None; out = f(); out = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = f()
        out = None
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for N in [int(v) for v in sys.argv[1:]] or (250, 1000):
    K, C = 7, 128
    step = N // 2
    Wt = max(1, round(1792 / N))
    T = step * (Wt + 1)
    W = (T - N) // step + 1
    R = int(1000 * 1024 / T)
    x = torch.randn((T, R, C), device=dev)
    tap = torch.randn((K, N), device=dev)
    for dbg in ("3", "0"):
        row = []
        for det in (None, "constant", "linear"):
            _lib.set_debug_env("SC_MTFFT_MIXED", "1")
            _lib.set_debug_env("SC_MTFFT_DEBUG", dbg)
            row.append(timed(lambda: engine.multitaper_spectra(x, tap, N, step, N, W, det)))
        print(f"N={N:5d} SC_MTFFT_DEBUG={dbg}: detrend None {row[0] * 1e3:.2f} ms, constant {row[1] * 1e3:.2f}, linear {row[2] * 1e3:.2f}", flush=True)
    _lib.set_debug_env("SC_MTFFT_DEBUG", None)
    del x
