"""Wall time from a NumPy time series to a NumPy result through the public API at the cfg3 shape (second call of each:
the first pays for page-locked buffers, the side stream and the allocator's first blocks)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spectral_connectivity_amd as sc
x32 = np.random.default_rng(3).standard_normal((1024, 1000, 128)).astype(np.float32)
x64 = x32.astype(np.float64)
kw = dict(sampling_frequency=1000.0, time_halfbandwidth_product=4, n_time_samples_per_window=256, n_time_samples_per_step=128)
for label, x, dtype in (("float32 input, complex64", x32, np.complex64), ("float64 input, complex64", x64, np.complex64),
                        ("float64 input, complex128 (default)", x64, np.complex128)):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m = sc.Multitaper(x, **kw)
        t1 = time.perf_counter()
        c = sc.Connectivity.from_multitaper(m, dtype=dtype)
        t2 = time.perf_counter()
        coh = c.coherence_magnitude()
        t3 = time.perf_counter()
        w = c.weighted_phase_lag_index()
        t4 = time.perf_counter()
    print(f"{label}: Multitaper() {1e3*(t1-t0):.1f} ms, from_multitaper {1e3*(t2-t1):.1f} ms, coherence_magnitude() "
          f"{1e3*(t3-t2):.1f} ms, weighted_phase_lag_index() {1e3*(t4-t3):.1f} ms, total {1e3*(t4-t0):.1f} ms")

# the torch-free host (ctypes + NumPy over sc_device_alloc / sc_memcpy_* / sc_stream_*) cannot share a process with the
# PyTorch host (two HIP runtimes): timed in a child process
import subprocess
code = r"""
import time, numpy as np
from spectral_connectivity_amd.numpy_host import NumpyHost
host = NumpyHost()
x32 = np.random.default_rng(3).standard_normal((1024, 1000, 128)).astype(np.float32)
kw = dict(sampling_frequency=1000.0, time_halfbandwidth_product=4, n_time_samples_per_window=256, n_time_samples_per_step=128)
for rep in range(3):
    t0 = time.perf_counter()
    out = host.connectivity(x32, measures=("coherence_magnitude", "weighted_phase_lag_index"), **kw)
    t1 = time.perf_counter()
print(f"NumPy host (no torch), float32 input: connectivity(coherence_magnitude + weighted_phase_lag_index) {1e3*(t1-t0):.1f} ms")
"""
out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                     cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print(out.stdout.strip() or out.stderr[-1500:])
