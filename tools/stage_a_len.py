"""Stage A alone at one window length (for rocprofv3 passes): STAGE_A_N samples, cfg3 data volume, STAGE_A_PLANES=1 for the planes
output; three launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine      # noqa: E402

dev = torch.device("cuda:0")
N = int(os.environ.get("STAGE_A_N", "250"))
K, C = 7, 128
step = N // 2
Wt = max(1, round(1792 / N))
T = step * (Wt + 1)
W = (T - N) // step + 1
R = int(1000 * 1024 / T)
x = torch.randn((T, R, C), device=dev)
tap = torch.randn((K, N), device=dev)
hint = (_lib.PLANE_CSM | _lib.PLANE_ABS_IM) if os.environ.get("STAGE_A_PLANES") == "1" else None
for _ in range(3):
    out = engine.multitaper_spectra(x, tap, N, step, N, W, "constant", planes_hint=hint)
    torch.cuda.synchronize()
    out = None
