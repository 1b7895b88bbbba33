"""Stage A, planes output against the complex64 output per window length (library timers, cfg3 volume).
    python tools/stage_a_planes_lengths.py [N ...]   (SC_MTFFT_DEBUG=32: without super-tiles)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine      # noqa: E402

dev = torch.device("cuda:0")
PL = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
lib = _lib.load()


def timed_lib(f, reps=5):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    lib.sc_timing_enable(1)
    _lib.last_timing()
    ts = []
    for _ in range(reps):
        f()
        torch.cuda.synchronize()
        ts.append(dict(_lib.last_timing()).get("mtfft_fused", float("nan")))
    lib.sc_timing_enable(0)
    return float(np.median(ts))


print("#     N | complex64 | planes | planes without super-tiles (SC_MTFFT_DEBUG=32)   ms")
for N in [int(v) for v in sys.argv[1:]] or (256, 512, 1024, 2048, 4096, 250, 500, 800, 1000, 1200):
    K, C = 7, 128
    step = N // 2
    Wt = max(1, round(1792 / N))
    T = step * (Wt + 1)
    W = (T - N) // step + 1
    R = int(1000 * 1024 / T)
    x = torch.randn((T, R, C), device=dev)
    tap = torch.randn((K, N), device=dev)
    row = []
    for hint, dbg in ((None, None), (PL, None), (PL, "32")):
        _lib.set_debug_env("SC_MTFFT_DEBUG", dbg)

        def f():
            sp = engine.multitaper_spectra(x, tap, N, step, N, W, "constant", planes_hint=hint)
            del sp
        row.append(timed_lib(f))
    _lib.set_debug_env("SC_MTFFT_DEBUG", None)
    print(f"N={N:5d} | " + " | ".join(f"{t:6.3f}" for t in row), flush=True)
    del x
