"""Profiling aid (GPU box): phase timers of the fused window/taper/FFT kernel (library rebuilt with -DMT_TRACE)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tools", "libsc_trace_mt.so")

if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    if "--build" in sys.argv:
        from spectral_connectivity_amd import _build
        _build.build(extra_flags=["-DMT_TRACE"], out=LIB)
        sys.exit(0)
    os.environ["SC_HIP_LIB"] = LIB
    import torch
    from spectral_connectivity_amd import _lib, engine
    x = torch.randn(1024, 1000, 128, device="cuda")
    tap = torch.randn(7, 256, device="cuda") * 0.01
    lib = ctypes.CDLL(LIB)
    names = ["load + detrend + samples to registers", "waits at the per-taper barrier (7 tapers)",
             "radix-16 passes (7 tapers)", "split + store issue (7 tapers)"]
    for label, hint in (("complex64 output", None), ("planes-format output", _lib.PLANE_CSM | _lib.PLANE_ABS_IM)):
        engine.multitaper_spectra(x, tap, 256, 128, 256, 7, "constant", planes_hint=hint)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 8)()
        lib.sc_debug_mtfft_trace(None, 1)
        engine.multitaper_spectra(x, tap, 256, 128, 256, 7, "constant", planes_hint=hint)
        torch.cuda.synchronize()
        lib.sc_debug_mtfft_trace(buf, 0)
        tot = sum(buf[i] for i in range(4))
        print("%s: shader-clock cycles of wave 0 of one workgroup (c-tile 0, trial 3, window 3); total %d" % (label, tot))
        for i, n in enumerate(names):
            print("  %-44s %8d  %5.1f %%" % (n, buf[i], 100.0 * buf[i] / max(tot, 1)))
