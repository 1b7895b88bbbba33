"""Profiling aid (GPU box): time stage B (fused CSM + |Im| kernel) with parts switched off (SC_FUSED_DEBUG)."""
import os
import subprocess
import sys

CODE = r'''
import torch
from spectral_connectivity_amd import engine, _lib
F, W, R, K, C = 129, 7, 1000, 7, 128
X = torch.randn(F, W, R, K, C, dtype=torch.complex64, device="cuda")
sp = engine.DeviceSpectra(X, (F, W, R, K, C), (W * R * K * C, R * K * C, K * C, C), 256, real_input=True)
planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
for _ in range(2):
    acc = engine.accumulate(sp, "trials_tapers", planes)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    acc = engine.accumulate(sp, "trials_tapers", planes)
b.record(); torch.cuda.synchronize()
print("%.3f ms" % (a.elapsed_time(b) / 5))
'''
for dbg, name in [(0, "full"), (1, "no CSM MFMAs"), (2, "no abs products"), (3, "staging only"),
                  (11, "staging only, no HBM loads"), (8, "no HBM loads")]:
    env = dict(os.environ, SC_FUSED_DEBUG=str(dbg))
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print("%-28s %s" % (name, out.stdout.strip() or out.stderr.strip()[-400:]))
