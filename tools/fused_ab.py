"""A/B of stage-B variants inside one process (same box, same clocks): SC_FUSED_DEBUG values given on the command line
(results are only right for the values documented as such in sc_fused.hip), cfg3 volume, alternating, median of 15."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine      # noqa: E402

variants = sys.argv[1:] or ["0", "64"]
dev = torch.device("cuda:0")
F, W, K = 129, 7, 7
for C in (128, 64, 160, 256):
    R = int(1000 * 128 / C)
    X = torch.view_as_complex(torch.randn((F, W, R, K, C, 2), dtype=torch.float32, device=dev))
    sp = engine.DeviceSpectra(X, (F, W, R, K, C), (W * R * K * C, R * K * C, K * C, C), 256, True, C_alloc=C)
    for name, planes in (("CSM+|Im|", _lib.PLANE_CSM | _lib.PLANE_ABS_IM), ("CSM", _lib.PLANE_CSM)):
        times = {v: [] for v in variants}
        for rep in range(17):
            for v in variants:
                _lib.set_debug_env("SC_FUSED_DEBUG", v)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = engine.accumulate(sp, "trials_tapers", planes)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                out = None
                if rep >= 2:
                    times[v].append(dt)
        print(f"C={C:4d} {name:9s}: " + "   ".join(f"dbg={v}: {np.median(times[v]) * 1e3:.3f} ms" for v in variants))
    del X, sp
_lib.set_debug_env("SC_FUSED_DEBUG", None)
