"""Build libsc_hip.so (hand-written HIP for gfx950 + rocFFT) in-tree with hipcc.

``python -m spectral_connectivity_amd._build`` or ``__graft_entry__.build()``.
hipcc cross-compiles for gfx950 without a GPU; the resulting .so sits next to this file so
that it travels with the source tree (no JIT cache, nothing in site-packages).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsc_hip.so")
SOURCES = ["sc_api.hip", "sc_taper.hip", "sc_mtfft.hip", "sc_csm.hip", "sc_nonlinear.hip", "sc_fused.hip", "sc_measure.hip",
           "sc_wilson.hip", "sc_wilson_fft.hip", "sc_mvar.hip", "sc_global.hip", "sc_canonical.hip"]
HEADERS = ["sc_common.h", "sc_stage.h", os.path.join("..", "..", "include", "sc_hip.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libsc_hip.so (set HIPCC=/path/to/hipcc)")


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 into spectral_connectivity_amd/libsc_hip.so."""
    if not force and not is_stale():
        return LIB
    cmd = [_hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
           "-Wno-unused-result", "-fno-slp-vectorize", *sources(), "-lrocfft", "-o", LIB + ".tmp"]
    if verbose:
        print("[spectral_connectivity_amd] " + " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
