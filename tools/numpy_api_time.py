"""The public classes on the torch-free host (SC_HIP_HOST=numpy) at the cfg3 shape: NumPy series in -> NumPy results out, wall time
per phase (third pass: page-locked buffers and device blocks are recycled from the second on).  Run as is: the switch is set here,
before the package is imported; torch is never imported."""
import os
import sys
import time

os.environ["SC_HIP_HOST"] = "numpy"
import numpy as np      # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spectral_connectivity_amd as sc      # noqa: E402

x32 = np.random.default_rng(3).standard_normal((1024, 1000, 128)).astype(np.float32)
kw = dict(sampling_frequency=1000.0, time_halfbandwidth_product=4, n_time_samples_per_window=256, n_time_samples_per_step=128)
for label, dtype in (("float32 engine (dtype=complex64)", np.complex64), ("float64 engine (dtype=complex128, the default)", np.complex128)):
    x = x32 if dtype == np.complex64 else x32.astype(np.float64)
    for rep in range(3):
        t0 = time.perf_counter()
        m = sc.Multitaper(x, **kw)
        c = sc.Connectivity.from_multitaper(m, dtype=dtype)
        t1 = time.perf_counter()
        coh = c.coherence_magnitude()
        t2 = time.perf_counter()
        w = c.weighted_phase_lag_index()
        t3 = time.perf_counter()
    print(f"{label}: constructors {1e3 * (t1 - t0):.1f} ms, coherence_magnitude() {1e3 * (t2 - t1):.1f} ms (upload + stages A, B, C + download), "
          f"weighted_phase_lag_index() {1e3 * (t3 - t2):.1f} ms, total {1e3 * (t3 - t0):.1f} ms; out {coh.shape} {coh.dtype} / {w.dtype}", flush=True)
assert "torch" not in sys.modules
print("torch imported:", "torch" in sys.modules)
