"""fp64 matrix-core CSM kernel alone at the cfg3 shape (SC_F64_OC: rows per staged chunk)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import engine, _lib
F, W, R, K, C = 129, 7, 1000, 7, 128
X = torch.randn(F, W, R, K, C, dtype=torch.complex128, device="cuda")
sp = engine.DeviceSpectra(X, (F, W, R, K, C), (W * R * K * C, R * K * C, K * C, C), 256, real_input=True)
for oc in ("8", "16", "8", "16"):
    os.environ["SC_F64_OC"] = oc
    for _ in range(2):
        engine.accumulate(sp, "trials_tapers", _lib.PLANE_CSM)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        engine.accumulate(sp, "trials_tapers", _lib.PLANE_CSM)
    b.record(); torch.cuda.synchronize()
    print(f"OC={oc}: CSM f64 {a.elapsed_time(b) / 5:.3f} ms")
