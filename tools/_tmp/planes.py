import sys, time, torch
sys.path.insert(0, "/root/repo")
from spectral_connectivity_amd import engine, _lib
dev = torch.device("cuda:0")
F, W, K = 129, 7, 7
cases = (("coherence", _lib.PLANE_CSM), ("wPLI", _lib.PLANE_CSM | _lib.PLANE_ABS_IM),
         ("debiased wPLI", _lib.PLANE_CSM | _lib.PLANE_ABS_IM | _lib.PLANE_IM_SQ), ("PLI", _lib.PLANE_SIGN_IM),
         ("PLV / PPC", _lib.PLANE_UNIT))
for C in (16, 64, 128):
    R = int(1000 * 128 / C)
    X = torch.view_as_complex(torch.randn((F, W, R, K, C, 2), dtype=torch.float32, device=dev))
    sp = engine.DeviceSpectra(X, (F, W, R, K, C), (W * R * K * C, R * K * C, K * C, C), 256, True)
    for tag, planes in cases:
        f = lambda: engine.accumulate(sp, "trials_tapers", planes)
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): a = f()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"C={C:3d} R={R:5d} {tag:14s}: {dt*1e3:7.2f} ms")
    del X, sp
