"""Stage B of the float64 engine at the cfg3 volume, kernel by kernel: the matrix-core CSM alone (three- and four-product
form, 4- and 8-wave workgroups), the |Im s| plane alone, and both (forked) -- library hipEvent timers, median of 5."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine      # noqa: E402

dev = torch.device("cuda:0")
C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
F, W, R, K = 129, 7, 1000 * 128 // C, 7
g = torch.Generator(device=dev).manual_seed(1)
X = torch.view_as_complex(torch.randn((F, W, R, K, C, 2), dtype=torch.float64, device=dev, generator=g))
sp = engine.DeviceSpectra(X, (F, W, R, K, C), (W * R * K * C, R * K * C, K * C, C), 256, True, C_alloc=C)
_lib.timing_enable(True)


def run(label, planes, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    _lib.reload_debug_env()
    ts = []
    for rep in range(7):
        _lib.last_timing()
        accum, _ = engine.accumulate(sp, "trials_tapers", planes)
        torch.cuda.synchronize()
        t = sum(ms for name, ms in _lib.last_timing() if name == "accumulate_f64")
        if rep >= 2:
            ts.append(t)
        del accum
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    _lib.reload_debug_env()
    print(f"C={C} {label:58s} {np.median(ts):7.3f} ms")


CSM, ABS = _lib.PLANE_CSM, _lib.PLANE_ABS_IM
print("# tools/f64_csm_time.py: sc_accumulate_f64 on complex128 spectra of the cfg3 volume")
run("CSM, three products, 8-wave workgroups (default)", CSM, {})
run("CSM, three products, 4-wave workgroups x 2 per CU", CSM, {"SC_F64_CSM3_WAVES": "4"})
run("CSM, four products (rounds 2-3)", CSM, {"SC_F64_FOUR_PRODUCTS": "1"})
run("|Im s| plane alone", CSM | ABS, {"SC_F64_WHICH_ABS_ONLY": "1"}) if False else None
run("CSM + |Im s| (forked), three products", CSM | ABS, {})
run("CSM + |Im s| (forked), three products, 4-wave", CSM | ABS, {"SC_F64_CSM3_WAVES": "4"})
run("CSM + |Im s| (forked), four products", CSM | ABS, {"SC_F64_FOUR_PRODUCTS": "1"})
run("CSM + |Im s| (not forked), three products", CSM | ABS, {"SC_F64_NO_FORK": "1"})
for S in (3, 4, 6, 7, 8):
    run(f"CSM, three products, 8-wave, split {S}", CSM, {"SC_F64_SPLIT": str(S)})
