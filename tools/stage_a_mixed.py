"""Stage A for the window lengths that are not powers of two (sc_mtfft_mixed.hip): correctness against the float64 transform and time
at the cfg3 data volume, for both geometries of every length (SC_MTFFT_MIXED_GEO), both outputs, and the round-2 kernel beside them.
    python tools/stage_a_mixed.py check      # small shapes, every length, against float64
    python tools/stage_a_mixed.py time       # cfg3 volume: ms and TB/s of spectra
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine      # noqa: E402

dev = torch.device("cuda:0")
GEOS = tuple(os.environ.get("MIX_GEOS", "0,1").split(","))
PL = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
LENGTHS = (100, 150, 160, 200, 240, 250, 300, 320, 360, 400, 450, 480, 500, 600, 750, 800, 900, 1000, 1200, 1250, 1500, 1600, 1800, 2000)


def env(**kw):
    for k, v in kw.items():
        _lib.set_debug_env(k, v)


def check():
    os.environ["SC_PLANES_MIN_CHANNELS"] = "2"
    os.environ["SC_PLANES_MIN_BYTES"] = "0"
    worst = 0.0
    for N in LENGTHS:
        for (C, L, detr) in ((6, N, "constant"), (34, N - 7, "linear"), (70, N, None)):
            T, R, K = 2 * N + 11, 3, 3
            step = N // 2
            W = (T - L) // step + 1
            g = torch.Generator(device=dev).manual_seed(N + C)
            x = torch.randn((T, R, C), device=dev, generator=g) + 3.0
            x[:, :, 0] *= 200.0
            x[:, :, C - 1] *= 2e-3
            h = torch.randn((K, L), device=dev, generator=g) / 30.0
            X64 = engine.multitaper_spectra_f64(x.double(), h.double(), L, step, N, W, detr).X
            amax = X64.abs().amax(dim=(0, 1, 2, 3))
            for geo in GEOS:
                env(SC_MTFFT_MIXED="1", SC_MTFFT_MIXED_GEO=geo)
                X = engine.multitaper_spectra(x, h, L, step, N, W, detr).X
                e1 = ((X - X64).abs().amax(dim=(0, 1, 2, 3)) / amax).max().item()
                e2 = float("nan")
                if C % 2 == 0:
                    sp = engine.multitaper_spectra(x, h, L, step, N, W, detr, planes_hint=PL)
                    assert sp.P is not None, "planes format expected"
                    e2 = ((sp.X - X64).abs().amax(dim=(0, 1, 2, 3)) / amax).max().item()
                env(SC_MTFFT_MIXED="0")
                Xo = engine.multitaper_spectra(x, h, L, step, N, W, detr).X
                e0 = ((Xo - X64).abs().amax(dim=(0, 1, 2, 3)) / amax).max().item()
                worst = max(worst, e1, e2 if e2 == e2 else 0.0)
                flag = "" if max(e1, e2 if e2 == e2 else 0.0) < 2.5e-6 else "   <-- FAIL"
                print(f"N={N:5d} C={C:3d} L={L:5d} {str(detr):8s} geo={geo}: c64 {e1:.2e}  planes {e2:.2e}  (round-2 kernel {e0:.2e}){flag}", flush=True)
    env(SC_MTFFT_MIXED=None, SC_MTFFT_MIXED_GEO=None)
    print("worst", worst)
    return worst < 2.5e-6


def timed(f, reps=5):
    out = f(); out = None; out = f(); out = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = f()
        out = None
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def bench(lengths):
    os.environ["SC_PLANES_MIN_BYTES"] = "0"
    print("# 128 channels, 7 tapers, half-overlapping windows, cfg3 data volume; ms (TB/s of spectra written)")
    print("#     N  W     R |  round-2 kernel | " + " | ".join(f"  geo {g}: c64   planes" for g in GEOS))
    for N in lengths:
        K, C = 7, 128
        step = N // 2
        Wt = max(1, round(1792 / N) )
        T = step * (Wt + 1)
        W = (T - N) // step + 1
        R = int(1000 * 1024 / T)
        x = torch.randn((T, R, C), device=dev)
        tap = torch.randn((K, N), device=dev)
        gb = (N // 2 + 1) * W * R * K * C * 8 / 1e9
        row = []
        env(SC_MTFFT_MIXED="0")
        row.append(timed(lambda: engine.multitaper_spectra(x, tap, N, step, N, W, "constant")))
        for geo in GEOS:
            env(SC_MTFFT_MIXED="1", SC_MTFFT_MIXED_GEO=geo)
            row.append(timed(lambda: engine.multitaper_spectra(x, tap, N, step, N, W, "constant")))
            sp = engine.multitaper_spectra(x, tap, N, step, N, W, "constant", planes_hint=PL)
            assert sp.P is not None
            del sp
            _lib.load().sc_timing_enable(1)
            ts = []
            for _ in range(5):
                sp = engine.multitaper_spectra(x, tap, N, step, N, W, "constant", planes_hint=PL)
                del sp
                torch.cuda.synchronize()
                ts.append(dict(_lib.last_timing()).get("mtfft_fused", float("nan")) * 1e-3)
            _lib.load().sc_timing_enable(0)
            row.append(float(np.median(ts)))
        print(f"N={N:5d} {W:2d} {R:5d} | " + " | ".join(f"{t * 1e3:6.2f} ({gb / t / 1e3:4.2f})" for t in row) + f"   [{gb:.2f} GB]", flush=True)
        del x
    env(SC_MTFFT_MIXED=None, SC_MTFFT_MIXED_GEO=None)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "check"
    if what == "check":
        sys.exit(0 if check() else 1)
    bench([int(v) for v in sys.argv[2:]] or LENGTHS)
