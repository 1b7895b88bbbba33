"""Breakdown of the torch-free host (numpy_host.NumpyHost) at the cfg3 shape: upload, stage A, stage B, epilogue,
download -- wall time per step with a stream synchronisation after each (third repetition)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib
from spectral_connectivity_amd.numpy_host import NumpyHost, MEASURES, PinnedArray
from spectral_connectivity_amd.transforms import Multitaper
host = NumpyHost()
x32 = np.random.default_rng(3).standard_normal((1024, 1000, 128)).astype(np.float32)
kw = dict(sampling_frequency=1000.0, time_halfbandwidth_product=4, n_time_samples_per_window=256, n_time_samples_per_step=128)
names = ("coherence_magnitude", "weighted_phase_lag_index")
_up, _al = host.upload, host.alloc
spent = {"upload": 0.0, "alloc": 0.0}
def upload(a):
    t0 = time.perf_counter(); r = _up(a); spent["upload"] += time.perf_counter() - t0; return r
def alloc(n):
    t0 = time.perf_counter(); r = _al(n); spent["alloc"] += time.perf_counter() - t0; return r
host.upload, host.alloc = upload, alloc
for rep in range(3):
    spent["upload"] = spent["alloc"] = 0.0
    t = [time.perf_counter()]
    m = Multitaper(x32, **kw); t.append(time.perf_counter())
    m.tapers; print(f"tapers {1e3 * (time.perf_counter() - t[-1]):.1f} ms")
    planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
    sp = host.spectra(m, planes_hint=planes); host.synchronize(); t.append(time.perf_counter())
    print(f"inside spectra(): uploads {1e3 * spent['upload']:.1f} ms (incl. their allocations), allocations {1e3 * spent['alloc']:.1f} ms")
    accum, n_bins, n_obs = host.accumulate(sp, "trials_tapers", planes); host.synchronize(); t.append(time.perf_counter())
    outs = []
    for name in names:
        dev = host.alloc(n_bins * 128 * 128 * 8)
        _lib.check(host.lib.sc_measure_f64(accum.ptr, n_bins, 128, planes, n_obs, MEASURES[name], dev.ptr, host.stream), "m")
        host.synchronize(); t.append(time.perf_counter())
        outs.append(host.download(dev, (7, 129, 128, 128), np.float64)); t.append(time.perf_counter())
        dev.free()
    for key in ("X", "P", "scale"):
        if sp.get(key) is not None:
            sp[key].free()
    accum.free()
    t.append(time.perf_counter())
lab = ["Multitaper()", "upload + stage A", "stage B", "epilogue 1", "download 1 (118 MB f64)", "epilogue 2", "download 2", "frees"]
print("NumPy host, cfg3, float32 input: " + ", ".join(f"{l} {1e3*(b-a):.1f} ms" for l, a, b in zip(lab, t[:-1], t[1:]))
      + f", total {1e3*(t[-1]-t[0]):.1f} ms")
# copy rates
src = PinnedArray.empty(host.lib, x32.shape, np.float32); src[:] = x32
for label, a in (("pageable", x32), ("page-locked", src)):
    d = host.upload(a); host.synchronize()
    t0 = time.perf_counter(); d = host.upload(a); host.synchronize(); t1 = time.perf_counter()
    print(f"h2d {label}: {a.nbytes / (t1 - t0) / 1e9:.1f} GB/s")
t0 = time.perf_counter(); back = host.download(d, x32.shape, np.float32); t1 = time.perf_counter()
t2 = time.perf_counter(); back = host.download(d, x32.shape, np.float32); t3 = time.perf_counter()
print(f"d2h into page-locked memory: first {x32.nbytes / (t1 - t0) / 1e9:.1f} GB/s, again {x32.nbytes / (t3 - t2) / 1e9:.1f} GB/s")
from ctypes import c_void_p
for rep in range(3):
    dst = np.empty(x32.shape, np.float32)
    t0 = time.perf_counter()
    _lib.check(host.lib.sc_memcpy_d2h(dst.ctypes.data_as(c_void_p), d.ptr, dst.nbytes, host.stream), "d2h"); host.synchronize()
    t1 = time.perf_counter()
    print(f"d2h into a fresh pageable array: {x32.nbytes / (t1 - t0) / 1e9:.1f} GB/s")
t0 = time.perf_counter()
_lib.check(host.lib.sc_memcpy_d2h(dst.ctypes.data_as(c_void_p), d.ptr, dst.nbytes, host.stream), "d2h"); host.synchronize()
print(f"d2h into a touched pageable array: {x32.nbytes / (time.perf_counter() - t0) / 1e9:.1f} GB/s")
t0 = time.perf_counter()
_lib.check(host.lib.sc_memcpy_d2h(back.ctypes.data_as(c_void_p), d.ptr, back.nbytes, host.stream), "d2h"); host.synchronize()
print(f"d2h into an existing page-locked array: {x32.nbytes / (time.perf_counter() - t0) / 1e9:.1f} GB/s")
for n in (118 << 20, 118 << 20, 1 << 30, 1 << 30):
    t0 = time.perf_counter(); b = host.alloc(n); host.synchronize(); t1 = time.perf_counter(); b.free(); host.synchronize(); t2 = time.perf_counter()
    print(f"sc_device_alloc {n >> 20} MB: {1e3 * (t1 - t0):.2f} ms, free {1e3 * (t2 - t1):.2f} ms")
for n in (118 << 20, 118 << 20):
    t0 = time.perf_counter(); a = PinnedArray.empty(host.lib, (n,), np.uint8); t1 = time.perf_counter(); del a; t2 = time.perf_counter()
    print(f"sc_host_alloc {n >> 20} MB: {1e3 * (t1 - t0):.2f} ms, free {1e3 * (t2 - t1):.2f} ms")
