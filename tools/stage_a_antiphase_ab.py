"""Stage A, complex64 output, at the cfg3 data volume (128 channels, 1000 trials, 7 tapers): the round-1..3 kernels (SC_MTFFT_LONG=0)
against the anti-phase kernel of sc_mtfft_long.hip from 256 samples on (SC_MTFFT_LONG=256), with either workgroup size
(SC_MTFFT_DEBUG=64: the other one); same process, alternating, two rounds."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine      # noqa: E402

C, K = 128, 7
VARIANTS = (("round-3", "0", None), ("anti-phase", "256", None), ("anti-phase, other size", "256", "64"))
for (T, L, step, R) in ((1024, 256, 128, 1000), (1024, 256, 256, 1000), (1024, 512, 256, 1000), (1024, 1024, 1024, 1000)):
    W = (T - L) // step + 1
    x = torch.randn(T, R, C, device="cuda")
    h = torch.randn(K, L, device="cuda") / L
    gb = 8.0 * (L // 2 + 1) * W * R * K * C / 1e9
    out, ms = {}, {}
    for rnd in range(2):
        for name, long_, dbg in VARIANTS:
            _lib.set_debug_env("SC_MTFFT_LONG", long_)
            _lib.set_debug_env("SC_MTFFT_DEBUG", dbg)
            for _ in range(2):
                sp = engine.multitaper_spectra(x, h, L, step, L, W, "constant")
            out[name] = sp.X.clone()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                sp = engine.multitaper_spectra(x, h, L, step, L, W, "constant")
            b.record()
            torch.cuda.synchronize()
            ms.setdefault(name, []).append(a.elapsed_time(b) / 10)
            del sp
    ref = out["round-3"]
    line = "   ".join(f"{n}: {min(v):.3f} ms ({gb / min(v):.2f} TB/s, diff {(out[n] - ref).abs().max().item() / ref.abs().max().item():.0e})"
                      for n, v in ms.items())
    print(f"L={L} step={step} W={W}: {line}")
    del out, ref, x
_lib.set_debug_env("SC_MTFFT_LONG", None)
_lib.set_debug_env("SC_MTFFT_DEBUG", None)
