"""Time the batched 2x2 Wilson factorisation (sc_wilson_factor_f64) at the cfg4 shape -- 2016 problems of
4096 bins -- with the fused causal-FFT kernel and with SC_WILSON_FFT=rocfft.  Spectra are built on the device
from random stable filters, so no host transfer is inside the timed region.  Usage: python tools/wilson_time.py [P N]"""
import ctypes
import os
import sys
import time
from ctypes import byref

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib                      # noqa: E402
from spectral_connectivity_amd.engine import _ptr, _stream      # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2016
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
lib = _lib.load()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
z = torch.exp(-2j * torch.pi * torch.arange(N, device=dev, dtype=torch.float64) / N)
B1 = torch.randn((P, 2, 2), generator=g, device=dev, dtype=torch.float64) * 0.3
B2 = torch.randn((P, 2, 2), generator=g, device=dev, dtype=torch.float64) * 0.12
F = (torch.eye(2, device=dev, dtype=torch.complex128)[None, None] + B1[:, None].to(torch.complex128) * z[None, :, None, None]
     + B2[:, None].to(torch.complex128) * (z * z)[None, :, None, None])
Sm = F @ F.conj().transpose(-1, -2)                              # (P, N, 2, 2)
S = torch.stack([Sm[..., 0, 0].real, Sm[..., 1, 1].real, Sm[..., 0, 1].real, Sm[..., 0, 1].imag], dim=1).contiguous()
del F, Sm
nbytes = ctypes.c_size_t()
_lib.check(lib.sc_granger_workspace_bytes(1, P, N, byref(nbytes)), "ws")
work = torch.empty((nbytes.value,), dtype=torch.uint8, device=dev)
G = torch.empty((P, 4, N), dtype=torch.complex128, device=dev)
n_iter = torch.empty((P,), dtype=torch.int32, device=dev)
status = torch.empty((P,), dtype=torch.int32, device=dev)
summary = (ctypes.c_int32 * 3)(0, 0, 0)


def run():
    _lib.check(lib.sc_wilson_factor_f64(_ptr(S), P, N, 1e-8, 60, _ptr(work), nbytes.value, _ptr(G), _ptr(n_iter),
                                        _ptr(status), summary, _stream()), "wilson")
    torch.cuda.synchronize()


res = {}
for mode in ("fused", "rocfft", "fused", "rocfft"):
    # (the library reads its switches once, at load: set_debug_env changes the variable AND has it read again)
    _lib.set_debug_env("SC_WILSON_FFT", "rocfft" if mode == "rocfft" else None)
    run()
    t0 = time.perf_counter()
    for _ in range(3):
        run()
    dt = (time.perf_counter() - t0) / 3
    res[mode] = G.clone()
    print(f"{mode:7s} P={P} N={N}: {dt * 1e3:8.2f} ms per factorisation, {summary[0]} iterations "
          f"({dt * 1e3 / max(summary[0], 1):.3f} ms each), not converged {summary[1]}, "
          f"mean iterations per problem {n_iter.float().mean().item():.1f}")
print("max |G_fused - G_rocfft| =", (res["fused"] - res["rocfft"]).abs().max().item())
