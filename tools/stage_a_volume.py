"""Stage A at 2048 / 4096 samples against the number of windows and the volume (the round-5 sweep showed 1.97 TB/s at N = 4096 with
three half-overlapping windows x 250 trials, 11 GB of spectra, against 3.0 TB/s at 5.5 GB with one window): which of the two it is."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import engine      # noqa: E402

dev = torch.device("cuda:0")


def timed(f, reps=4):
    out = f(); out = None; out = f(); out = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = f()
        out = None
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


print("#     N  W    R  step |     ms   GB of spectra   TB/s")
for N in (2048, 4096):
    for (W, R, step) in ((1, 250, N), (1, 750, N), (3, 83, N // 2), (3, 250, N // 2), (3, 250, N), (2, 375, N // 2)):
        K, C = 7, 128
        T = N + (W - 1) * step
        x = torch.randn((T, R, C), device=dev)
        tap = torch.randn((K, N), device=dev)
        dt = timed(lambda: engine.multitaper_spectra(x, tap, N, step, N, W, "constant"))
        gb = (N // 2 + 1) * W * R * K * C * 8 / 1e9
        print(f"N={N:5d} {W:2d} {R:4d} {step:5d} | {dt * 1e3:6.2f}   {gb:6.2f}   {gb / dt / 1e3:.2f}", flush=True)
        del x
