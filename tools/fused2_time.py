"""Time of stage B on the planes format (sc_fused2.hip) at the cfg3 volume under the SC_FUSED_DEBUG ablation switches
(1 = CSM waves skip their MFMAs, 2 = |Im s| waves skip their products, 8 = no HBM loads after the first chunk; results are
wrong when set), next to the complex64 kernel.  Alternating inside one process, median of 15."""
import os
import sys
import time
from ctypes import byref, c_double

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine      # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
variants = sys.argv[1:] or ["0", "1", "2", "3", "8", "11"]
F, W, K = 129, 7, 7
for C in (128, 64):
    R = int(1000 * 128 / C)
    X = torch.view_as_complex(torch.randn((F, W, R, K, C, 2), dtype=torch.float32, device=dev))
    sp = engine.DeviceSpectra(X, (F, W, R, K, C), (W * R * K * C, R * K * C, K * C, C), 256, True, C_alloc=C)
    d = sp.desc("trials_tapers")
    rb = lib.sc_planes_row_bytes(C)
    P = torch.empty((F * W * R * K * rb,), dtype=torch.uint8, device=dev)
    scale = torch.empty((2 * C,), dtype=torch.float32, device=dev)
    work = torch.empty((C,), dtype=torch.int32, device=dev)
    _lib.check(lib.sc_planes_scales_from_spectra_f32(X.data_ptr(), F * W * R * K, C, scale.data_ptr(), work.data_ptr(), None), "scales")
    _lib.check(lib.sc_planes_from_spectra_f32(X.data_ptr(), byref(d), scale.data_ptr(), P.data_ptr(), None), "to planes")
    n_bins, fpb, _, _ = engine.accum_layout(sp, "trials_tapers", planes)
    ws_bytes = int(lib.sc_fused_workspace_bytes(byref(d), planes))
    ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=dev)
    out = torch.empty((n_bins, fpb), dtype=torch.float32, device=dev)
    times = {("new", v): [] for v in variants}
    times[("old", "0")] = []
    clocks = {k: [] for k in times}
    for rep in range(17):
        for key in times:
            _lib.set_debug_env("SC_FUSED_DEBUG", key[1])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if key[0] == "new":
                _lib.check(lib.sc_fused2_csm_absim_f32(P.data_ptr(), byref(d), scale.data_ptr(), planes, out.data_ptr(), ws.data_ptr(), ws_bytes, None), "fused2")
            else:
                engine.accumulate(sp, "trials_tapers", planes)
            torch.cuda.synchronize()
            if rep >= 2:
                times[key].append(time.perf_counter() - t0)
                if key[0] == "new":
                    ghz = c_double(0.0)
                    lib.sc_debug_fused2_clock(byref(ghz))       # sustained shader clock of that launch (in-kernel counters)
                    clocks[key].append(ghz.value)
    print(f"C={C:4d}: " + "   ".join(f"{k[0]} dbg={k[1]}: {np.median(v) * 1e3:.3f} ms" + (f" @{np.median(clocks[k]):.2f} GHz" if clocks[k] else "")
                                     for k, v in times.items()))
_lib.set_debug_env("SC_FUSED_DEBUG", None)
