"""PLV plane of the float64 engine at the cfg3 shape."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine
F, W, R, K, C = 129, 7, 1000, 7, 128
X = torch.view_as_complex(torch.randn((F, W, R, K, C, 2), dtype=torch.float64, device="cuda"))
sp = engine.DeviceSpectra(X, (F, W, R, K, C), (W * R * K * C, R * K * C, K * C, C), 256, True)
for name, planes in (("CSM", _lib.PLANE_CSM), ("unit (PLV)", _lib.PLANE_UNIT), ("sign (PLI)", _lib.PLANE_SIGN_IM)):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        engine.accumulate(sp, "trials_tapers", planes)
        torch.cuda.synchronize()
    print(f"float64 engine, {name}: {1e3 * (time.perf_counter() - t0):.1f} ms")
