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

# where the call's time goes: the library's stage timers (hipEvents around every entry point) against the wall clock
from spectral_connectivity_amd import _lib      # noqa: E402
_lib.timing_enable(True)
c = sc.Connectivity.from_multitaper(m)
torch.cuda.synchronize()
_lib.last_timing()
t0 = time.perf_counter()
G = c._mvar_factor_device()
torch.cuda.synchronize()
t1 = time.perf_counter()
from spectral_connectivity_amd import engine      # noqa: E402
d = engine.mvar_measure(G, _lib.MVAR_DTF)
torch.cuda.synchronize()
t2 = time.perf_counter()
h = engine.to_host(d)
t3 = time.perf_counter()
print(f"  factor {1e3 * (t1 - t0):.1f} ms (records + Wilson), measure {1e3 * (t2 - t1):.1f} ms, download of {h.nbytes / 1e6:.0f} MB {1e3 * (t3 - t2):.1f} ms")
print("  library timers:", ", ".join(f"{k} {v:.2f}" for k, v in _lib.last_timing()))
