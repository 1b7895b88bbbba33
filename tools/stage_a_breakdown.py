"""Stage A per entry point (library hipEvent timers) for window lengths off the fused kernel's list and for long windows."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine      # noqa: E402

dev = torch.device("cuda:0")
for (T, L, R) in ((1000, 250, 1000), (1000, 200, 1000), (3000, 1000, 300), (2048, 1024, 1000), (4096, 2048, 250), (8192, 4096, 250)):
    step, K, C = L // 2, 7, 128
    x = torch.randn((T, R, C), device=dev)
    tap = torch.randn((K, L), device=dev)
    W = (T - L) // step + 1
    for _ in range(2):
        sp = engine.multitaper_spectra(x, tap, L, step, L, W, "constant")
        del sp
    torch.cuda.synchronize()
    _lib.timing_enable(True)
    for _ in range(3):
        sp = engine.multitaper_spectra(x, tap, L, step, L, W, "constant")
        del sp
    torch.cuda.synchronize()
    t = _lib.last_timing()
    _lib.timing_enable(False)
    agg = {}
    for n, ms in t:
        agg[n] = agg.get(n, 0.0) + ms / 3
    gb = (L // 2 + 1) * W * R * K * C * 8 / 1e9
    print(f"N={L:5d} W={W} R={R}: " + ", ".join(f"{k} {v:.2f} ms" for k, v in agg.items()) + f"  ({gb:.2f} GB of spectra)")
    del x
