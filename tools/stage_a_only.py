"""Stage A alone at the cfg3 shape (for rocprofv3 --pmc passes over mtfft16_kernel): three launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import engine      # noqa: E402

dev = torch.device("cuda:0")
T, L, R, K, C = 1024, 256, 1000, 7, 128
x = torch.randn((T, R, C), device=dev)
tap = torch.randn((K, L), device=dev)
W = (T - L) // (L // 2) + 1
for _ in range(3):
    out = engine.multitaper_spectra(x, tap, L, L // 2, L, W, "constant")
    torch.cuda.synchronize()
    out = None
