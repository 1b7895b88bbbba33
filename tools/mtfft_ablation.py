"""Profiling aid (GPU box): time the fused window/taper/FFT kernel with parts switched off (SC_MTFFT_DEBUG)."""
import os
import sys
import subprocess

CODE = r'''
import torch
from spectral_connectivity_amd import engine
x = torch.randn(1024, 1000, 128, device="cuda")
tap = torch.randn(7, 256, device="cuda")
for _ in range(3):
    sp = engine.multitaper_spectra(x, tap, 256, 128, 256, 7, "constant")
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    sp = engine.multitaper_spectra(x, tap, 256, 128, 256, 7, "constant")
b.record(); torch.cuda.synchronize()
print("%.3f ms" % (a.elapsed_time(b) / 10))
'''
for dbg, name in [(0, "full"), (1, "no HBM stores"), (2, "no radix-16 passes"), (4, "no split/store loop"),
                  (3, "no passes, no stores"), (6, "load + detrend only")]:
    env = dict(os.environ, SC_MTFFT_DEBUG=str(dbg))
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print("%-24s %s" % (name, out.stdout.strip() or out.stderr.strip()[-300:]))
