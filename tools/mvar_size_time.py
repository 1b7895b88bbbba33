"""Full-system Wilson factorisation + one directed measure across system sizes (float64 records), ms per call."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spectral_connectivity_amd as sc      # noqa: E402

for C, T, R in ((64, 256, 40), (128, 256, 60), (130, 256, 60), (160, 256, 70), (192, 256, 80), (256, 256, 100)):
    rng = np.random.default_rng(C)
    e = rng.standard_normal((T + 8, R, C))
    x = e.copy()
    for t in range(2, T + 8):
        x[t] += 0.35 * x[t - 1] - 0.2 * x[t - 2]
        x[t, :, 1:] += 0.25 * x[t - 1, :, :-1]
    x = x[8:]
    c = sc.Connectivity.from_multitaper(sc.Multitaper(x, sampling_frequency=128.0, time_halfbandwidth_product=3),
                                        dtype=np.complex128)
    c.coherence_magnitude()                       # records on the device
    t0 = time.perf_counter()
    d = c.directed_transfer_function()
    t1 = time.perf_counter()
    w = c._last_wilson
    c2 = sc.Connectivity.from_multitaper(sc.Multitaper(x, sampling_frequency=128.0, time_halfbandwidth_product=3),
                                         dtype=np.complex128)
    c2.coherence_magnitude()
    t2 = time.perf_counter()
    c2.directed_transfer_function()
    t3 = time.perf_counter()
    print(f"C={C:4d} N={T}: first call {1e3 * (t1 - t0):8.1f} ms, second object {1e3 * (t3 - t2):8.1f} ms, "
          f"{w['iterations']} iterations, not converged {w['not_converged']}, finite {bool(np.isfinite(d).all())}")
