"""Stage B per plane family at the cfg3 data volume for the channel counts the matrix-core kernel serves (ms per accumulate())."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine      # noqa: E402

F, W, K = 129, 7, 7
fams = (("CSM", _lib.PLANE_CSM), ("CSM+|Im|", _lib.PLANE_CSM | _lib.PLANE_ABS_IM),
        ("CSM+|Im|+Im^2", _lib.PLANE_CSM | _lib.PLANE_ABS_IM | _lib.PLANE_IM_SQ), ("sign Im", _lib.PLANE_SIGN_IM))
print("channels  " + "  ".join(f"{n:>16s}" for n, _ in fams) + "   (ms; per-plane VALU kernel for Im^2 / sign in brackets)")
for C in (60, 64, 96, 128):
    R = int(1000 * 128 / C)
    X = torch.view_as_complex(torch.randn((F, W, R, K, C, 2), dtype=torch.float32, device="cuda"))
    sp = engine.DeviceSpectra(X, (F, W, R, K, C), (W * R * K * C, R * K * C, K * C, C), 256, True)

    def timed(f, reps=5):
        f(); f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    row = [timed(lambda: engine.accumulate(sp, "trials_tapers", pl)) for _, pl in fams]
    old = [timed(lambda: engine.accumulate(sp, "trials_tapers", pl, use_fused=False), reps=2) for _, pl in fams[2:]]
    print(f"{C:8d}  " + "  ".join(f"{v:16.2f}" for v in row) + f"   [{old[0]:.1f}, {old[1]:.1f} with every plane on its own kernel]")
    del X, sp
