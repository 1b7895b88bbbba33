"""Stage A at 2048 samples: the waves of a transform meeting at an LDS counter (default since round 6) against the workgroup-barrier
form (SC_MTFFT_DEBUG=256), both outputs, library timers, cfg3 volume."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine      # noqa: E402

dev = torch.device("cuda:0")
PL = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
lib = _lib.load()


def timed_lib(f, reps=7):
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


N, K, C = 2048, 7, 128
for (W, R, step) in ((1, 500, N), (3, 250, N // 2)):
    T = N + (W - 1) * step
    x = torch.randn((T, R, C), device=dev)
    tap = torch.randn((K, N), device=dev)
    gb = (N // 2 + 1) * W * R * K * C * 8 / 1e9
    for dbg in ("256", None):
        _lib.set_debug_env("SC_MTFFT_DEBUG", dbg)
        row = []
        for hint in (None, PL):
            def f():
                sp = engine.multitaper_spectra(x, tap, N, step, N, W, "constant", planes_hint=hint)
                del sp
            row.append(timed_lib(f))
        print(f"N={N} W={W} R={R} ({gb:.2f} GB) {'barriers' if dbg else 'counters'}: complex64 {row[0]:.3f} ms ({gb / row[0]:.2f} TB/s), planes {row[1]:.3f} ms ({gb / row[1]:.2f} TB/s)", flush=True)
    _lib.set_debug_env("SC_MTFFT_DEBUG", None)
    del x
