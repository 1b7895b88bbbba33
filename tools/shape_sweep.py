"""Stage A and stage B across shapes at the cfg3 data volume (6.5 GB of spectra): window lengths 64..4096 and
non-power-of-two lengths for stage A; 2..256 channels and every accumulator plane family for stage B.
Wall-clock per call over repeated launches with the device synchronised (inputs resident in HBM)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine      # noqa: E402

dev = torch.device("cuda:0")


def timed(f, reps=3):
    out = f(); out = None; out = f(); out = None      # one output alive at a time: the caching allocator reuses its block
    torch.cuda.synchronize()                           # (a second multi-GB block would add a one-off 0.2 s hipMalloc)
    t0 = time.perf_counter()
    for _ in range(reps):
        out = f()
        out = None
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


print("# stage A: 128 channels, 7 tapers, half-overlapping windows; output bytes / time")
for (T, L, R) in () if os.environ.get("SC_SWEEP_STAGE_B_ONLY") else ((1024, 64, 1000), (1024, 128, 1000), (1024, 256, 1000), (1024, 512, 1000), (2048, 1024, 1000),
                  (4096, 2048, 250), (8192, 4096, 250), (1000, 250, 1000), (1000, 200, 1000), (3000, 1000, 300)):
    step, K, C = L // 2, 7, 128
    x = torch.randn((T, R, C), device=dev)
    tap = torch.randn((K, L), device=dev)
    W = (T - L) // step + 1
    dt = timed(lambda: engine.multitaper_spectra(x, tap, L, step, L, W, "constant"))
    gb = (L // 2 + 1) * W * R * K * C * 8 / 1e9
    print(f"N={L:5d} W={W:2d} R={R:4d}: {dt * 1e3:7.2f} ms, {gb:5.2f} GB of spectra -> {gb / dt / 1e3:.2f} TB/s")
    del x

print("# stage B: 129 bins x 7 windows, trials x 7 tapers per bin scaled so that the spectra are 6.47 GB; ms per accumulate()")
F, W, K = 129, 7, 7
families = (("CSM (coherence)", _lib.PLANE_CSM), ("CSM+|Im| (wPLI)", _lib.PLANE_CSM | _lib.PLANE_ABS_IM),
            ("+Im^2 (debiased wPLI)", _lib.PLANE_CSM | _lib.PLANE_ABS_IM | _lib.PLANE_IM_SQ),
            ("sign Im (PLI)", _lib.PLANE_SIGN_IM), ("s/|s| (PLV, PPC)", _lib.PLANE_UNIT))
print("channels  " + "  ".join(f"{n:>22s}" for n, _ in families))
CHANNELS = (2, 4, 8, 16, 19, 32, 40, 44, 48, 50, 52, 56, 64, 96, 128, 130, 160, 192, 224, 256)
if os.environ.get("SC_SWEEP_CHANNELS"):
    CHANNELS = tuple(int(v) for v in os.environ["SC_SWEEP_CHANNELS"].split(","))
for C in CHANNELS:
    R = max(4, int(1000 * 128 / C))
    x = torch.randn((8, R, C), device=dev)                    # only to build spectra of the right (padded) layout
    Cp = C + 1 if (C % 2 and C + 1 <= 128) else C
    X = torch.view_as_complex(torch.randn((F, W, R, K, Cp, 2), dtype=torch.float32, device=dev))
    if Cp != C:
        X[..., C] = 0
    sp = engine.DeviceSpectra(X, (F, W, R, K, C), (W * R * K * Cp, R * K * Cp, K * Cp, Cp), 256, True, C_alloc=Cp)
    row = []
    for _, planes in families:
        row.append(timed(lambda: engine.accumulate(sp, "trials_tapers", planes)) * 1e3)
    print(f"{C:8d}  " + "  ".join(f"{v:22.2f}" for v in row))
    if Cp % 2 == 0:
        # the same spectra in the planes format (two f16 pieces per real number, sc_fused2.hip): what the float32 engine runs from
        # 49 channels on for these families (s/|s| has no planes-format kernel)
        from ctypes import byref
        lib = _lib.load()
        d = sp.desc("trials_tapers", padded=True)
        P = torch.zeros((F * W * R * K * lib.sc_planes_row_bytes(Cp),), dtype=torch.uint8, device=dev)
        scale = torch.empty((2 * Cp,), dtype=torch.float32, device=dev)
        work = torch.empty((Cp,), dtype=torch.int32, device=dev)
        _lib.check(lib.sc_planes_scales_from_spectra_f32(X.data_ptr(), F * W * R * K, Cp, scale.data_ptr(), work.data_ptr(), None), "scales")
        _lib.check(lib.sc_planes_from_spectra_f32(X.data_ptr(), byref(d), scale.data_ptr(), P.data_ptr(), None), "to planes")
        spp = engine.DeviceSpectra(None, (F, W, R, K, C), sp.strides, 256, True, C_alloc=Cp, P=P, scale=scale)
        row = []
        for _, planes in families[:4]:
            row.append(timed(lambda: engine.accumulate(spp, "trials_tapers", planes)) * 1e3)
        print(f"  planes  " + "  ".join(f"{v:22.2f}" for v in row))
        del P, spp
    del X, sp, x
