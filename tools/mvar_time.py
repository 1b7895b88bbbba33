"""Full C x C Wilson factorisation + DTF (up to 256 channels): 40 trials x 512 samples, N = 512 two-sided bins.  Usage: python tools/mvar_time.py [C T window].  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spectral_connectivity_amd as sc      # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
L = int(sys.argv[3]) if len(sys.argv) > 3 else None      # window length (default: one window of T samples)
rng = np.random.default_rng(9)
R = 40
e = rng.standard_normal((T + 100, R, C))
x = np.zeros_like(e)
for t in range(2, T + 100):
    x[t] = 0.45 * x[t - 1] - 0.25 * x[t - 2] + e[t]
    x[t, :, 1:] += 0.3 * x[t - 1, :, :-1]
x = x[100:].astype(np.float32)
kw = dict(n_time_samples_per_window=L) if L else {}
m = sc.Multitaper(x, sampling_frequency=500.0, time_halfbandwidth_product=3, **kw)
for rep in range(2):
    c = sc.Connectivity.from_multitaper(m)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dtf = c.directed_transfer_function()
    torch.cuda.synchronize()
    print(f"C={C} T={T} window={L or T}: directed_transfer_function() {1e3 * (time.perf_counter() - t0):.1f} ms, "
          f"Wilson iterations {c._last_wilson['iterations']}, not converged {c._last_wilson['not_converged']}, out {dtf.shape}")
