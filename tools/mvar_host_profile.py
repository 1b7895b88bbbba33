"""Host-side profile (cProfile) of one directed_transfer_function() call on a fresh Connectivity: where the wall time beyond the
library's own timers goes.  Usage: python tools/mvar_host_profile.py [C T window]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spectral_connectivity_amd as sc      # noqa: E402
from spectral_connectivity_amd import _lib      # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1792
L = int(sys.argv[3]) if len(sys.argv) > 3 else 256
rng = np.random.default_rng(9)
x = rng.standard_normal((T, 40, C)).astype(np.float32)
x[1:] += 0.5 * x[:-1]
m = sc.Multitaper(x, sampling_frequency=500.0, time_halfbandwidth_product=3, n_time_samples_per_window=L)
sc.Connectivity.from_multitaper(m).directed_transfer_function()
_lib.timing_enable(True)
for rep in range(2):
    c = sc.Connectivity.from_multitaper(m)
    torch.cuda.synchronize()
    _lib.last_timing()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    d = c.directed_transfer_function()
    pr.disable()
    print(f"wall {1e3 * (time.perf_counter() - t0):.1f} ms; library timers:", ", ".join(f"{k} {v:.2f}" for k, v in _lib.last_timing()))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
