"""Profiling aid (GPU box): per-wave phase timers of the fused stage-B kernel (library rebuilt with -DFU_TRACE)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tools", "libsc_trace.so")


def build():
    sys.path.insert(0, ROOT)
    from spectral_connectivity_amd import _build
    _build.build(extra_flags=["-DFU_TRACE"], out=LIB)


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
        sys.exit(0)
    os.environ["SC_HIP_LIB"] = LIB
    os.environ["SC_FUSED_SPLIT"] = "1"          # one workgroup per bin: workgroup 0 sees every chunk (set BEFORE the library is
                                                # loaded below: its switches are a snapshot taken at load)
    sys.path.insert(0, ROOT)
    import torch
    from spectral_connectivity_amd import engine, _lib
    F, W, R, K, C = 129, 7, 1000, 7, 128
    X = torch.randn(F, W, R, K, C, dtype=torch.complex64, device="cuda")
    sp = engine.DeviceSpectra(X, (F, W, R, K, C), (W * R * K * C, R * K * C, K * C, C), 256, real_input=True)
    planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
    engine.accumulate(sp, "trials_tapers", planes)
    torch.cuda.synchronize()
    lib = ctypes.CDLL(LIB)
    buf = (ctypes.c_ulonglong * 48)()
    lib.sc_debug_fused_trace(None, 1)
    engine.accumulate(sp, "trials_tapers", planes)
    torch.cuda.synchronize()
    lib.sc_debug_fused_trace(buf, 0)
    n_chunks = (R * K + 31) // 32
    print("cycles per chunk (workgroup 0, %d chunks); CSM waves: [-, products, flush, barrier]; "
          "abs waves: [staging, products, -, barrier]" % n_chunks)
    for w in range(12):
        v = [buf[w * 4 + i] / n_chunks for i in range(4)]
        print("wave %2d %s  %s   total %.0f" % (w, "CSM" if w < 4 else "abs", "  ".join("%8.0f" % x for x in v), sum(v)))
