"""Where the wall time of the public-API pass goes (series resident in HBM, cfg3): host phases by perf_counter, device stages by the
library's timers, the clock stage B sustained; then the same pass with the downloads left out (results kept on the device) and a
back-to-back loop of passes to see what the clock does when the GPU is never idle."""
import os
import sys
import time
from ctypes import byref, c_double

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spectral_connectivity_amd as sc      # noqa: E402
from spectral_connectivity_amd import _lib, engine      # noqa: E402

dev = torch.device("cuda:0")
T, R, C = 1024, 1000, 128
x = torch.randn((T, R, C), device=dev)
kw = dict(sampling_frequency=1000.0, time_halfbandwidth_product=4.0, n_time_samples_per_window=256, n_time_samples_per_step=128)
lib = _lib.load()
lib.sc_timing_enable(1)


def clock():
    ck = c_double(0.0)
    lib.sc_debug_fused2_clock(byref(ck))
    return ck.value


for rep in range(5):
    torch.cuda.synchronize()
    _lib.last_timing()
    t0 = time.perf_counter()
    m = sc.Multitaper(x, **kw)
    t1 = time.perf_counter()
    c = sc.Connectivity.from_multitaper(m, dtype=np.complex64)
    t2 = time.perf_counter()
    coh = c.coherence_magnitude()
    t3 = time.perf_counter()
    w = c.weighted_phase_lag_index()
    t4 = time.perf_counter()
    st = {}
    for name, ms in _lib.last_timing():
        st[name] = st.get(name, 0.0) + ms
    print(f"pass {rep}: Multitaper() {1e3 * (t1 - t0):.2f}  from_multitaper {1e3 * (t2 - t1):.2f}  coherence_magnitude() {1e3 * (t3 - t2):.2f}  "
          f"weighted_phase_lag_index() {1e3 * (t4 - t3):.2f}  total {1e3 * (t4 - t0):.2f} ms | device " +
          ", ".join(f"{k} {v:.2f}" for k, v in st.items()) + f" | stage B clock {clock():.2f} GHz", flush=True)
    del m, c, coh, w

# the same device work without the public classes' host phases and downloads, from an idle GPU (one pass, then idle 20 ms)
from spectral_connectivity_amd.transforms import _make_tapers      # noqa: E402
tapers = _make_tapers(256, 1000.0, 4.0, 7)
h = torch.from_numpy(np.ascontiguousarray(tapers.T / 1000.0, dtype=np.float32)).to(dev)
PL = _lib.PLANE_CSM | _lib.PLANE_ABS_IM


def chain():
    sp = engine.multitaper_spectra(x, h, 256, 128, 256, 7, "constant", planes_hint=PL)
    accum, n_obs = engine.accumulate(sp, "trials_tapers", PL, fold=False)
    del sp
    return engine.measure_multi(accum, C, PL, n_obs, [_lib.M_COHERENCE_MAGNITUDE, _lib.M_WPLI])


for idle_ms in (0, 2, 5, 20):
    cl, tt = [], []
    for rep in range(6):
        torch.cuda.synchronize()
        time.sleep(idle_ms * 1e-3)
        t0 = time.perf_counter()
        out = chain()
        torch.cuda.synchronize()
        tt.append(1e3 * (time.perf_counter() - t0))
        cl.append(clock())
        del out
    print(f"engine chain after {idle_ms:2d} ms of idle GPU: {np.median(tt):.2f} ms per pass, stage B clock {np.median(cl):.2f} GHz", flush=True)
