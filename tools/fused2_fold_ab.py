"""Fold interval of stage B's two-level summation (SC_FUSED_FOLD_OBS) at the full cfg3 size: stage-B time (library timers, median)
and the accuracy of coherence / wPLI against the float64 reference (tests/fp64_device_ref.py), in units of the full-depth bound
3e-6 |ref| + 2e-7 max|ref| of tests/test_gpu_full_depth.py.  One process, one box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fp64_device_ref import measures_fp64, spectra_fp64, sums_fp64      # noqa: E402
from spectral_connectivity_amd import _lib, engine                      # noqa: E402
from spectral_connectivity_amd.transforms import _make_tapers           # noqa: E402

FS = 1000.0
T, R, C, L, step, NW = 1024, 1000, 128, 256, 128, 4
W = (T - L) // step + 1
rng = np.random.default_rng(3)
x = rng.standard_normal((T, R, C)).astype(np.float32)
t = np.arange(T) / FS
x += (0.5 * np.sin(2 * np.pi * 60.0 * t[:, None, None] + 2 * np.pi * np.arange(C)[None, None, :] / C)).astype(np.float32)
tapers = _make_tapers(L, FS, NW, 7)
X = spectra_fp64(x, tapers, FS, L, step, L)
csm, ab = sums_fp64(X)
n_obs = X.shape[2] * X.shape[3]
del X
ref = measures_fp64(csm, ab, n_obs)
del csm, ab
dev = torch.device("cuda:0")
xd = torch.from_numpy(x).to(dev)
h = torch.from_numpy(np.ascontiguousarray(tapers.T / FS, dtype=np.float32)).to(dev)
planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
sp = engine.multitaper_spectra(xd, h, L, step, L, W, "constant", planes_hint=planes)
assert sp.P is not None
_lib.timing_enable(True)


def worst(got, name):
    b = ref[name]
    a = got.reshape(b.shape).cpu().numpy().astype(np.float64)
    ok = ~np.isnan(b)
    return float((np.abs(a[ok] - b[ok]) / (3e-6 * np.abs(b[ok]) + 2e-7 * np.abs(b[ok]).max())).max())


print("# tools/fused2_fold_ab.py: observations between the folds of a tile's f32 accumulators into the record (three parts of 2333 per bin)")
print(f"{'fold every':>12s} {'stage B ms':>11s} {'coherence err/bound':>20s} {'wPLI err/bound':>15s}")
for fold in (512, 1024, 2048, 1 << 20):
    _lib.set_debug_env("SC_FUSED_FOLD_OBS", fold)
    ts = []
    for rep in range(11):
        _lib.last_timing()
        accum, n = engine.accumulate(sp, "trials_tapers", planes, fold=False)
        torch.cuda.synchronize()
        ts.append(sum(ms for name, ms in _lib.last_timing() if name == "fused_stage_b"))
    coh, wpli = engine.measure_multi(accum, C, planes, n, [_lib.M_COHERENCE_MAGNITUDE, _lib.M_WPLI])
    label = "never" if fold >= (1 << 20) else str(fold)
    print(f"{label:>12s} {np.median(ts[2:]):11.3f} {worst(coh, 'coherence_magnitude'):20.2f} {worst(wpli, 'weighted_phase_lag_index'):15.2f}")
_lib.set_debug_env("SC_FUSED_FOLD_OBS", None)
