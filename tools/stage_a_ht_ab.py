"""Stage A at 256 / 512 samples with either workgroup size of the anti-phase kernel (SC_MTFFT_DEBUG=64 takes the other one), both outputs:
what the planes output costs beside the complex64 output (library timers, cfg3 volume)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from spectral_connectivity_amd import _lib, engine
dev = torch.device("cuda:0"); PL = _lib.PLANE_CSM | _lib.PLANE_ABS_IM; lib = _lib.load()
def timed_lib(f, reps=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); lib.sc_timing_enable(1); _lib.last_timing(); ts = []
    for _ in range(reps):
        f(); torch.cuda.synchronize(); ts.append(dict(_lib.last_timing()).get("mtfft_fused", float("nan")))
    lib.sc_timing_enable(0); return float(np.median(ts))
for N in (256, 512):
    K, C = 7, 128; step = N // 2; Wt = max(1, round(1792 / N)); T = step * (Wt + 1); W = (T - N) // step + 1; R = int(1000 * 1024 / T)
    x = torch.randn((T, R, C), device=dev); tap = torch.randn((K, N), device=dev)
    for dbg in (None, "64"):
        _lib.set_debug_env("SC_MTFFT_DEBUG", dbg)
        row = []
        for hint in (None, PL):
            def f():
                sp = engine.multitaper_spectra(x, tap, N, step, N, W, "constant", planes_hint=hint); del sp
            row.append(timed_lib(f))
        print(f"N={N} SC_MTFFT_DEBUG={dbg}: complex64 {row[0]:.3f} ms, planes {row[1]:.3f} ms", flush=True)
    _lib.set_debug_env("SC_MTFFT_DEBUG", None)
