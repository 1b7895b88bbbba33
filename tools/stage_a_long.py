"""Stage A for long power-of-two windows at the cfg3 data volume (128 channels, 7 tapers, one window per trial): the transposed-series /
anti-phase kernel of sc_mtfft_long.hip (default) against the round-3 kernels (SC_MTFFT_LONG=0), same process, alternating; checks
the two against each other.  `python tools/stage_a_long.py [N ...]`; under rocprofv3 --kernel-trace --stats it gives the per-kernel split."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine      # noqa: E402

C, K = 128, 7
sizes = [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096]
variants = os.environ.get("SC_LONG_VARIANTS", "0,1").split(",")
for N in sizes:
    R = 1024 * 1000 // N
    x = torch.randn(N, R, C, device="cuda")
    h = torch.randn(K, N, device="cuda") / N
    gb = (4.0 * N * R * C + 8.0 * (N // 2 + 1) * R * K * C) / 1e9
    out, ms = {}, {}
    for v in variants:
        _lib.set_debug_env("SC_MTFFT_LONG", v)
        for _ in range(2):
            sp = engine.multitaper_spectra(x, h, N, N, N, 1, "linear")
        out[v] = sp.X.clone()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            sp = engine.multitaper_spectra(x, h, N, N, N, 1, "linear")
        b.record(); torch.cuda.synchronize()
        ms[v] = a.elapsed_time(b) / 10
        del sp
    ref = out[variants[0]]
    scale = ref.abs().max().item()
    line = "   ".join(f"SC_MTFFT_LONG={v}: {ms[v]:6.3f} ms {gb / ms[v]:5.2f} TB/s" for v in variants)
    diff = max((out[v] - ref).abs().max().item() for v in variants) / scale
    print(f"N={N:5d} R={R:4d}: {line}   max |diff| / max |X| = {diff:.1e}")
    del out, ref, x
_lib.set_debug_env("SC_MTFFT_LONG", None)
