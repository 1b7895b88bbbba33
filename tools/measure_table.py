"""Every expectation-type measure of the Connectivity API at the cfg3 shape, both engines: device time of the call (library
hipEvent timers: stage B passes + epilogue; stage A is shared and listed once) and wall time including the copy of the
float64 result to the host."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spectral_connectivity_amd as sc      # noqa: E402
from spectral_connectivity_amd import _lib   # noqa: E402

rng = np.random.default_rng(3)
x = rng.standard_normal((1024, 1000, 128)).astype(np.float32)
kw = dict(sampling_frequency=1000.0, time_halfbandwidth_product=4, n_time_samples_per_window=256, n_time_samples_per_step=128)
names = ["power", "coherency", "coherence_magnitude", "coherence_phase", "imaginary_coherence", "phase_locking_value",
         "pairwise_phase_consistency", "phase_lag_index", "debiased_squared_phase_lag_index", "weighted_phase_lag_index",
         "debiased_squared_weighted_phase_lag_index"]
_lib.timing_enable(True)
for dtype in (np.complex64, np.complex128):
    m = sc.Multitaper(x, **kw)
    print(f"# dtype={np.dtype(dtype).name}")
    for name in names:
        c = sc.Connectivity.from_multitaper(m, dtype=dtype)       # fresh: no cached records
        c.power() if name != "power" else None                    # stage A (and the CSM record) outside this row
        # twice, each on a fresh object: the first call of a record / result size pays the caching allocators' first
        # hipMalloc / page-locking of that block (round 2's 71 ms wPLI "outlier"); the second is the steady state
        walls = []
        for rep in range(2):
            if rep:
                c = sc.Connectivity.from_multitaper(m, dtype=dtype)
                c.power() if name != "power" else None
            torch.cuda.synchronize()
            _lib.last_timing()
            t0 = time.perf_counter()
            out = getattr(c, name)()
            walls.append(time.perf_counter() - t0)
            dev = sum(ms for _, ms in _lib.last_timing())
            del out
            out = None
        print(f"{name:45s} device {dev:8.2f} ms   call {1e3 * walls[1]:8.1f} ms (first call of this size {1e3 * walls[0]:6.1f} ms)")
