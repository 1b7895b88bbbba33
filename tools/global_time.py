"""Global coherence at the cfg5 shape (256 ch x 500 trials x 1024 samples: one window, 1024 two-sided bins) and at 128 ch."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spectral_connectivity_amd as sc      # noqa: E402

for C, R in ((128, 500), (256, 500)):
    x = np.random.default_rng(C).standard_normal((1024, R, C)).astype(np.float32)
    x[:, :, : C // 2] += np.random.default_rng(1).standard_normal((1024, R, 1)).astype(np.float32)
    m = sc.Multitaper(x, sampling_frequency=1000.0, time_halfbandwidth_product=3)
    c = sc.Connectivity.from_multitaper(m, dtype=np.complex64)
    c.coherence_magnitude()                    # stages A and B outside the timed region
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        vals, vecs = c.global_coherence(max_rank=2)
        torch.cuda.synchronize()
        print(f"C={C}: global_coherence(max_rank=2) over {vals.shape[1]} bins {1e3 * (time.perf_counter() - t0):.1f} ms")

# canonical coherence with large groups at the same shape: 4 groups of 64 and 2 groups of 128 channels (513 bins)
x = np.random.default_rng(7).standard_normal((1024, 500, 256)).astype(np.float32)
x += (0.6 * np.repeat(np.random.default_rng(8).standard_normal((1024, 500, 4)), 64, axis=2)).astype(np.float32)
m = sc.Multitaper(x, sampling_frequency=1000.0, time_halfbandwidth_product=3)
c = sc.Connectivity.from_multitaper(m, dtype=np.complex64)
c.coherence_magnitude()
# ((32,) * 8: the wave-per-problem kernel of groups up to 32; (33,) + (32,) * 6 + (31,): the same work on the workgroup-per-problem kernel)
for sizes in ((16,) * 16, (32,) * 8, (33,) + (32,) * 6 + (31,), (24,) * 10 + (16,), (64,) * 4, (128,) * 2):
    labels = np.repeat(np.arange(len(sizes)), sizes)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cc, _ = c.canonical_coherence(labels)
        torch.cuda.synchronize()
    print(f"canonical coherence, {len(sizes)} groups of {sizes[0]} ({sizes[-1]}): {1e3 * (time.perf_counter() - t0):.1f} ms, out {cc.shape}")
