"""Canonical coherence at the cfg5 shape (256 channels in 16 groups of 16, 500 trials x 1024 samples, 513 bins x 120
group pairs).  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel times; prints the wall time of
repeated calls (device work + the 1 MB result copy)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spectral_connectivity_amd as sc      # noqa: E402

rng = np.random.default_rng(5)
C, T, R = 256, 1024, 500
x = (rng.standard_normal((T, R, C)) + 0.4 * rng.standard_normal((T, R, 1))).astype(np.float32)
labels = np.arange(C) // 16
m = sc.Multitaper(x, sampling_frequency=1000.0, time_halfbandwidth_product=3)
c = sc.Connectivity.from_multitaper(m)
cc, _ = c.canonical_coherence(labels)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    cc, _ = c.canonical_coherence(labels)
torch.cuda.synchronize()
print(f"canonical_coherence(): {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per call, output {cc.shape}, "
      f"range [{np.nanmin(cc):.3f}, {np.nanmax(cc):.3f}]")
