"""Epilogue alone at the cfg3 shape (coherence magnitude + wPLI from one record): median of 20 launches, ms."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine      # noqa: E402

dev = torch.device("cuda:0")
F, W, K, C, R = 129, 7, 7, 128, 40
X = torch.view_as_complex(torch.randn((F, W, R, K, C, 2), dtype=torch.float32, device=dev))
sp = engine.DeviceSpectra(X, (F, W, R, K, C), (W * R * K * C, R * K * C, K * C, C), 256, True, C_alloc=C)
planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
accum, n = engine.accumulate(sp, "trials_tapers", planes)
ts = []
for rep in range(24):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = engine.measure_multi(accum, C, planes, n, [_lib.M_COHERENCE_MAGNITUDE, _lib.M_WPLI])
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    if rep == 0:
        chk = [float(o.double().nan_to_num().abs().sum()) for o in out]
    out = None
print(f"measure_multi(coherence magnitude, wPLI): {np.median(ts[4:]) * 1e3:.3f} ms   checksums {chk}")
