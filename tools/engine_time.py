"""The float32 and the float64 engine side by side on the BASELINE configurations: stage A / stage B / epilogue
(coherence magnitude + wPLI) from the library's own hipEvent timers (sc_last_timing), milliseconds per pass."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine                     # noqa: E402
from spectral_connectivity_amd.transforms import dpss_windows          # noqa: E402

CONFIGS = {                                                            # BASELINE.json configs[1..4]
    "cfg2": dict(T=1024, R=100, C=32, NW=3.0, L=1024, step=1024),
    "cfg3": dict(T=1024, R=1000, C=128, NW=4.0, L=256, step=128),
    "cfg4": dict(T=4096, R=200, C=64, NW=4.0, L=4096, step=4096),
    "cfg5": dict(T=1024, R=500, C=256, NW=4.0, L=1024, step=1024),
}
FS = 1000.0


def one_pass(x, h, cfg, f64):
    L, step = cfg["L"], cfg["step"]
    W = int(np.floor(cfg["T"] / step - L / step + 1))
    planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
    if f64:
        sp = engine.multitaper_spectra_f64(x, h, L, step, L, W, "constant")
    else:
        sp = engine.multitaper_spectra(x, h, L, step, L, W, "constant", planes_hint=planes)
    accum, n_obs = engine.accumulate(sp, "trials_tapers", planes)
    del sp
    a = engine.measure(accum, cfg["C"], planes, n_obs, _lib.M_COHERENCE_MAGNITUDE, wide=f64)
    b = engine.measure(accum, cfg["C"], planes, n_obs, _lib.M_WPLI, wide=f64)
    return a, b


def main():
    dev = torch.device("cuda", 0)
    _lib.timing_enable(True)
    print("# tools/engine_time.py: ms per pass (library hipEvent timers, median of 5 after 2 warm-up passes)")
    only = sys.argv[1:]                                        # e.g. "cfg3 float64" restricts the table (profiling)
    for name, cfg in CONFIGS.items():
        if only and name not in only:
            continue
        K = int(2 * cfg["NW"] - 1)
        tap, _ = dpss_windows(cfg["L"], cfg["NW"], K, is_low_bias=False)
        tap = np.asarray(tap)
        if tap.shape[0] != K:
            tap = tap.T
        for f64 in (False, True):
            if only and not ({"float32", "float64"} & set(only)) <= {"float64" if f64 else "float32"}:
                continue
            real = torch.float64 if f64 else torch.float32
            h = torch.from_numpy(np.ascontiguousarray(tap * np.sqrt(FS) / FS)).to(dev, real)
            x = torch.randn((cfg["T"], cfg["R"], cfg["C"]), device=dev, dtype=real)
            rows = []
            for it in range(7):
                one_pass(x, h, cfg, f64)
                torch.cuda.synchronize()
                t = _lib.last_timing()
                if it >= 2:
                    rows.append(t)
            names = [n for n, _ in rows[0]]
            med = np.median(np.array([[ms for _, ms in r] for r in rows]), axis=0)
            total = float(med.sum())
            parts = ", ".join(f"{n} {m:.3f}" for n, m in zip(names, med))
            print(f"{name} {'float64' if f64 else 'float32'} engine: {total:8.3f} ms   ({parts})")
            del x, h
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
