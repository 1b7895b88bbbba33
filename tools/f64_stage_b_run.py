"""float64 engine, stage B at the cfg3 shape: a few launches of the fp64 matrix-core CSM kernel and of the fp64 |Im s| plane
kernel, for rocprofv3 (--kernel-trace --stats, or --pmc)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import engine, _lib
F, W, R, K, C = 129, 7, 1000, 7, 128
X = torch.randn(F, W, R, K, C, dtype=torch.complex128, device="cuda")
sp = engine.DeviceSpectra(X, (F, W, R, K, C), (W * R * K * C, R * K * C, K * C, C), 256, real_input=True)
_lib.set_debug_env("SC_F64_NO_FORK", "1")         # (read by the library at load: set and re-read)
for _ in range(3):
    engine.accumulate(sp, "trials_tapers", _lib.PLANE_CSM | _lib.PLANE_ABS_IM)
torch.cuda.synchronize()
