"""Stage A into the planes format under the SC_MTFFT_DEBUG switches (1 = no HBM stores, 4 = no split / store loop, 2 = no FFT
passes; results wrong when set), next to the complex64 output and the scale pre-pass alone; cfg3 shape, median of 15."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine      # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
T, R, C, L, step, K = 1024, 1000, 128, 256, 128, 7
W = (T - L) // step + 1
x = torch.randn((T, R, C), dtype=torch.float32, device=dev)
h = torch.randn((K, L), dtype=torch.float32, device=dev) * 0.01
planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
variants = [("complex64", None, "0"), ("planes", planes, "0"), ("planes, nt stores", planes, "32"), ("planes, no stores", planes, "1"), ("planes, no store loop", planes, "4"),
            ("complex64, no stores", None, "1"), ("complex64, no store loop", None, "4")]
if len(sys.argv) > 1 and sys.argv[1] == "quick":          # (under rocprofv3 --pmc: the two plain variants, three passes)
    variants = variants[:2]
ts = {v[0]: [] for v in variants}
N_REP = 5 if len(variants) == 2 else 17
ts["scales only"] = []
scale = torch.empty((2 * C,), dtype=torch.float32, device=dev)
work = torch.empty((C,), dtype=torch.int32, device=dev)
for rep in range(N_REP):
    for name, hint, dbg in variants:
        _lib.set_debug_env("SC_MTFFT_DEBUG", dbg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sp = engine.multitaper_spectra(x, h, L, step, L, W, "constant", planes_hint=hint)
        torch.cuda.synchronize()
        if rep >= 2:
            ts[name].append(time.perf_counter() - t0)
        del sp
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lib.sc_planes_scales_from_series_f32(x.data_ptr(), T, R, C, 1.0, scale.data_ptr(), work.data_ptr(), None)
    torch.cuda.synchronize()
    if rep >= 2:
        ts["scales only"].append(time.perf_counter() - t0)
_lib.set_debug_env("SC_MTFFT_DEBUG", None)
for k, v in ts.items():
    print(f"{k:28s} {np.median(v) * 1e3:.3f} ms")
