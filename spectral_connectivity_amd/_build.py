"""Build libsc_hip.so (hand-written HIP for gfx950 + rocFFT) in-tree with hipcc.

``python -m spectral_connectivity_amd._build`` or ``__graft_entry__.build()``.
hipcc cross-compiles for gfx950 without a GPU; the resulting .so sits next to this file so
that it travels with the source tree (no JIT cache, nothing in site-packages).

Every source is compiled to its own object under ``build/`` (in parallel, re-used while the source, the
headers and the flags are unchanged -- the key is a hash of their contents), then linked.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsc_hip.so")
OBJ_DIR = os.path.join(os.path.dirname(HERE), "build", "obj")
SOURCES = ["sc_api.hip", "sc_taper.hip", "sc_mtfft.hip", "sc_mtfft_long.hip", "sc_mtfft_mixed.hip", "sc_mtfft_f64.hip", "sc_csm.hip", "sc_nonlinear.hip", "sc_fused.hip", "sc_fused2.hip", "sc_measure.hip",
           "sc_wilson.hip", "sc_wilson_fft.hip", "sc_wilson_pair.hip", "sc_mvar.hip", "sc_global.hip", "sc_canonical.hip", "sc_f64.hip",
           "sc_timing.hip", "sc_memory.hip", "sc_comm.hip"]
HEADERS = ["sc_common.h", "sc_stage.h", "sc_fused_common.h", "sc_jacobi.h", "sc_wilson_fft.h", "sc_mtfft_bfly.h", os.path.join("..", "..", "include", "sc_hip.h")]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-unused-result", "-fno-slp-vectorize"]


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


def _digest(paths, extra):
    h = hashlib.sha256(" ".join(extra).encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:20]


def build(force=False, verbose=True, extra_flags=(), out=None):
    """Compile every HIP source for gfx950 into spectral_connectivity_amd/libsc_hip.so (or ``out``)."""
    out = out or LIB
    if not force and out == LIB and not is_stale():
        return LIB
    hipcc = _hipcc()
    flags = FLAGS + list(extra_flags)
    headers = [os.path.join(CSRC, h) for h in HEADERS if os.path.exists(os.path.join(CSRC, h))]
    if force and os.path.isdir(OBJ_DIR) and not extra_flags:
        shutil.rmtree(OBJ_DIR)                        # a forced build really compiles every source
    os.makedirs(OBJ_DIR, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(OBJ_DIR, f"{os.path.basename(src)}.{_digest([src] + headers, flags)}.o")
        if not os.path.exists(obj):
            cmd = [hipcc, *flags, "-c", src, "-o", obj + ".tmp"]
            if verbose:
                print("[spectral_connectivity_amd] " + " ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
            os.replace(obj + ".tmp", obj)
        return obj

    with ThreadPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 1))) as pool:
        objs = list(pool.map(compile_one, sources()))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-lrocfft", "-o", out + ".tmp"]
    if verbose:
        print("[spectral_connectivity_amd] " + " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(out + ".tmp", out)
    keep = set(objs)                                  # objects of superseded sources are dropped
    for name in os.listdir(OBJ_DIR):
        path = os.path.join(OBJ_DIR, name)
        if path not in keep and name.endswith(".o") and not extra_flags:
            os.remove(path)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
