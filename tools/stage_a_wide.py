"""Stage A (sc_multitaper_fft_f32) for long power-of-two windows at the cfg3 data volume, per workgroup width
(SC_MTFFT_WIDE: 0 = 256-thread workgroups, 1 = 512 at N=1024 / 1024 at N=2048, 2 = also 1024 at N=4096, 3 = 512 at 2048 / 4096);
checks every variant against the 256-thread result."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine

C, K = 128, 7
for N, R in ((1024, 1000), (2048, 500), (4096, 250), (512, 1000)):
    T = N * (1024 * 1000 // (N * R)) if N * R <= 1024 * 1000 else N
    W = T // N
    x = torch.randn(T, R, C, device="cuda")
    h = torch.randn(K, N, device="cuda") / N
    F = N // 2 + 1
    gb = (4.0 * T * R * C + 8.0 * F * W * R * K * C) / 1e9
    ref = None
    for wide in ("0", "1", "2", "3"):
        _lib.set_debug_env("SC_MTFFT_WIDE", wide)
        for det in ("constant", "constant", "linear"):
            sp = engine.multitaper_spectra(x, h, N, N, N, W, det)
            if det == "linear":
                lin = sp.X.clone()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            sp = engine.multitaper_spectra(x, h, N, N, N, W, "constant")
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        if ref is None:
            ref, ref_lin = sp.X.clone(), lin
            err = 0.0
        else:
            err = max((sp.X - ref).abs().max().item(), (lin - ref_lin).abs().max().item()) / ref.abs().max().item()
        print(f"N={N:5d} R={R:4d} W={W} wide={wide}: {ms:7.3f} ms  {gb / ms:6.2f} TB/s   max diff vs 256-thread {err:.1e}")
        del sp
    del ref, ref_lin, lin
