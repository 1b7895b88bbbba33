"""Ablation of sc_mtfft_mixed.hip at the cfg3 data volume: SC_MTFFT_DEBUG 0 (whole kernel), 1 (no split / store loop), 2 (no passes),
3 (prologue + slot barriers only).  Results are WRONG with a debug flag set; only the time is read."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine      # noqa: E402

dev = torch.device("cuda:0")
GEOS = tuple(os.environ.get("MIX_GEOS", "0,1").split(","))
PL = _lib.PLANE_CSM | _lib.PLANE_ABS_IM


def timed(f, reps=5):
    out = f(); out = None; out = f(); out = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = f()
        out = None
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


lengths = [int(v) for v in sys.argv[1:]] or [250, 1000]
print("#     N geo out |  whole | no store | no passes | neither   (ms)")
for N in lengths:
    K, C = 7, 128
    step = N // 2
    Wt = max(1, round(1792 / N))
    T = step * (Wt + 1)
    W = (T - N) // step + 1
    R = int(1000 * 1024 / T)
    x = torch.randn((T, R, C), device=dev)
    tap = torch.randn((K, N), device=dev)
    for geo in GEOS:
        for out in ("c64", "planes"):
            row = []
            for dbg in ("0", "1", "2", "3"):
                _lib.set_debug_env("SC_MTFFT_MIXED", "1")
                _lib.set_debug_env("SC_MTFFT_MIXED_GEO", geo)
                _lib.set_debug_env("SC_MTFFT_DEBUG", dbg)
                hint = PL if out == "planes" else None
                row.append(timed(lambda: engine.multitaper_spectra(x, tap, N, step, N, W, "constant", planes_hint=hint)))
            print(f"N={N:5d}  {geo}  {out:6s} | " + " | ".join(f"{t * 1e3:6.2f}" for t in row), flush=True)
    del x
for k in ("SC_MTFFT_MIXED", "SC_MTFFT_MIXED_GEO", "SC_MTFFT_DEBUG"):
    _lib.set_debug_env(k, None)
