"""Stage A straight into the planes format (sc_multitaper_fft_planes_f32) and into complex64, both against the float64
transform of the same samples: error per channel relative to the channel's largest coefficient, the scales chosen from the series, and the time of both
(one process, alternating, median of 15) at the cfg3 volume and a few other window lengths."""
import os
import sys
import time
from ctypes import byref

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine, transforms      # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")


def run(T, R, C, L, step, NW, timing=False, detrend="constant"):
    N = L
    K = int(2 * NW - 1)
    W = (T - L) // step + 1
    F = N // 2 + 1
    tapers = transforms.dpss_windows(L, NW, K)[0] if hasattr(transforms, "dpss_windows") else None
    h = torch.from_numpy(np.ascontiguousarray(np.asarray(tapers)[:K] * np.sqrt(1000.0) / 1000.0, dtype=np.float32)).to(dev)
    g = torch.Generator(device=dev).manual_seed(T + C)
    x = torch.randn((T, R, C), dtype=torch.float32, device=dev, generator=g)
    x += 0.5 * torch.sin(2 * np.pi * 60 * torch.arange(T, device=dev) / 1000.0)[:, None, None]
    x[:, :, C // 2] *= 1e-3                                   # a quiet channel
    x[:, :, 1] *= 300.0                                       # a loud one
    sp = engine.multitaper_spectra(x, h, L, step, N, W, detrend)
    X = sp.X
    rb = lib.sc_planes_row_bytes(C)
    P = torch.empty((F * W * R * K * rb,), dtype=torch.uint8, device=dev)
    scale = torch.empty((2 * C,), dtype=torch.float32, device=dev)
    work = torch.empty((C,), dtype=torch.int32, device=dev)
    hsum = float(h.abs().sum(dim=1).max().item())
    tw = engine.twiddles(N, dev)

    def planes():
        _lib.check(lib.sc_planes_scales_from_series_f32(x.data_ptr(), T, R, C, hsum, scale.data_ptr(), work.data_ptr(), None), "scales")
        _lib.check(lib.sc_multitaper_fft_planes_f32(x.data_ptr(), T, R, C, L, step, W, N, h.data_ptr(), K, _lib.DETREND[detrend],
                                                    tw.data_ptr(), scale.data_ptr(), P.data_ptr(), None), "stage A planes")
    planes()
    d = sp.desc("trials_tapers")
    Xb = torch.zeros_like(X)
    _lib.check(lib.sc_spectra_from_planes_f32(P.data_ptr(), byref(d), scale.data_ptr(), Xb.data_ptr(), None), "from planes")
    torch.cuda.synchronize()
    # error relative to the channel's largest coefficient (what the f32 transform's own rounding is relative to), against float64
    X64 = engine.multitaper_spectra_f64(x.double(), h.double(), L, step, N, W, detrend).X
    amax = X64.abs().amax(dim=(0, 1, 2, 3))
    e_pl = (Xb - X64).abs().amax(dim=(0, 1, 2, 3)) / amax
    e_c64 = (X - X64).abs().amax(dim=(0, 1, 2, 3)) / amax
    print(f"T={T} R={R} C={C} L={L} W={W} K={K}: max_ch |err| / max_ch|X|: planes {e_pl.max().item():.2e} (quiet ch {e_pl[C // 2].item():.1e}, "
          f"partner of the loud ch {e_pl[0].item():.1e}), complex64 {e_c64.max().item():.2e} (quiet {e_c64[C // 2].item():.1e}, partner {e_c64[0].item():.1e}); "
          f"scaled |X| max = {(X64.abs() * scale[:C]).max().item():.0f}; finite: {bool(torch.isfinite(Xb).all())}")
    del X64
    if timing:
        ts = {"complex64": [], "planes (+ scales)": []}
        for rep in range(17):
            for name, fn in (("complex64", lambda: engine.multitaper_spectra(x, h, L, step, N, W, detrend)), ("planes (+ scales)", planes)):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                if rep >= 2:
                    ts[name].append(time.perf_counter() - t0)
        print("    " + "   ".join(f"{k}: {np.median(v) * 1e3:.3f} ms" for k, v in ts.items()))


if __name__ == "__main__":
    run(1024, 20, 128, 256, 128, 4)
    run(1024, 20, 100, 256, 128, 4, detrend="linear")
    run(1024, 20, 64, 128, 64, 3)
    run(2048, 10, 32, 64, 64, 2)
    run(2048, 10, 40, 512, 256, 3)
    run(4096, 6, 48, 1024, 1024, 3)
    run(1024, 1000, 128, 256, 128, 4, timing=True)
