import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import spectral_connectivity_amd as sc
from spectral_connectivity_amd import _lib
L = int(sys.argv[1]); C = int(sys.argv[2]); R = int(sys.argv[3])
rng = np.random.default_rng(1)
x = rng.standard_normal((L, R, C)).astype(np.float32); x[1:] += 0.5 * x[:-1]; x[:, :, 1:] += 0.3 * x[:, :, :-1]
m = sc.Multitaper(x, sampling_frequency=1000.0, time_halfbandwidth_product=3)
_lib.timing_enable(True)
for rep in range(3):
    c = sc.Connectivity.from_multitaper(m); c.coherence_magnitude(); torch.cuda.synchronize(); _lib.last_timing()
    t0 = time.perf_counter(); g = c.pairwise_spectral_granger_prediction(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"L={L} C={C}: {1e3*dt:.1f} ms, timers:", ", ".join(f"{k} {v:.2f}" for k, v in _lib.last_timing()), c._last_wilson["iterations"])
