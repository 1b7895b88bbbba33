"""Stage B on the planes format (sc_fused2.hip) against the complex64 kernel (sc_fused.hip) on the same random spectra:
records compared plane by plane, round trip of the format conversion, and the time of both at the cfg3 volume."""
import os
import sys
import time
from ctypes import byref

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine      # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
PLANES0 = _lib.PLANE_CSM | _lib.PLANE_ABS_IM


def run(C, R, F=5, W=2, K=7, timing=False, planes=PLANES0):
    torch.manual_seed(C * 1000 + R)
    X = torch.view_as_complex(torch.randn((F, W, R, K, C, 2), dtype=torch.float32, device=dev))
    X = X * (0.5 + torch.rand((1, 1, 1, 1, C), device=dev)) + 0.3 * X[..., :1]          # correlated channels
    sp = engine.DeviceSpectra(X, (F, W, R, K, C), (W * R * K * C, R * K * C, K * C, C), 256, True, C_alloc=C)
    d = sp.desc("trials_tapers")
    rb = lib.sc_planes_row_bytes(C)
    P = torch.empty((F * W * R * K * rb,), dtype=torch.uint8, device=dev)
    scale = torch.empty((2 * C,), dtype=torch.float32, device=dev)
    work = torch.empty((C,), dtype=torch.int32, device=dev)
    _lib.check(lib.sc_planes_scales_from_spectra_f32(X.data_ptr(), F * W * R * K, C, scale.data_ptr(), work.data_ptr(), None), "scales")
    _lib.check(lib.sc_planes_from_spectra_f32(X.data_ptr(), byref(d), scale.data_ptr(), P.data_ptr(), None), "to planes")
    Xb = torch.zeros_like(X)
    _lib.check(lib.sc_spectra_from_planes_f32(P.data_ptr(), byref(d), scale.data_ptr(), Xb.data_ptr(), None), "from planes")
    torch.cuda.synchronize()
    rt = ((Xb - X).abs() / X.abs().clamp_min(1e-30)).max().item()
    print(f"    format round trip: max relative error {rt:.2e} (22 significant bits: <= 2.4e-7); scales 2^{torch.log2(scale[:C]).min().item():.0f} .. 2^{torch.log2(scale[:C]).max().item():.0f}")
    ok = lib.sc_fused2_supported(byref(d), planes)
    ref, n_obs = engine.accumulate(sp, "trials_tapers", planes)
    torch.cuda.synchronize()
    n_bins, fpb, _, _ = engine.accum_layout(sp, "trials_tapers", planes)
    if not ok:
        print(f"C={C} R={R}: not supported (n_obs={n_obs})")
        return
    ws_bytes = int(lib.sc_fused_workspace_bytes(byref(d), planes))
    ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=dev)
    out = torch.full((n_bins, fpb), float("nan"), dtype=torch.float32, device=dev)
    _lib.check(lib.sc_fused2_csm_absim_f32(P.data_ptr(), byref(d), scale.data_ptr(), planes, out.data_ptr(), ws.data_ptr(), ws_bytes, None), "fused2")
    torch.cuda.synchronize()
    # fp64 truth of the three planes for the scale of the errors
    Xd = X.to(torch.complex128)
    S = torch.einsum("fwrkc,fwrkd->wfcd", Xd, Xd.conj())                       # [W][F][C][C]

    npl = {_lib.PLANE_CSM: 2, PLANES0: 3, PLANES0 | _lib.PLANE_IM_SQ: 4, _lib.PLANE_SIGN_IM: 1}[planes]
    nt = fpb // npl // 256
    a, b = out.view(n_bins, npl, nt, 256), ref.view(n_bins, npl, nt, 256)
    NB = (C + 15) // 16
    smax = S.abs().amax().item()
    errs = []
    names = {1: ["sum sign Im s"], 2: ["Re S", "Im S"], 3: ["Re S", "Im S", "sum |Im s|"], 4: ["Re S", "Im S", "sum |Im s|", "sum (Im s)^2"]}[npl]
    for pl, name in enumerate(names):
        # only entries of real channels: compare tile by tile on the valid part
        worst = 0.0
        t = 0
        for bi in range(NB):
            for bj in range(bi, NB):
                ni, nj = min(16, C - 16 * bi), min(16, C - 16 * bj)
                ta = a[:, pl, t].view(n_bins, 16, 16)[:, :ni, :nj]
                tb = b[:, pl, t].view(n_bins, 16, 16)[:, :ni, :nj]
                if bi == bj:
                    iu = torch.triu_indices(ni, nj, offset=1, device=dev)   # (the diagonal of Im-type planes is rounding noise in either kernel; no measure reads it)
                    ta, tb = ta[:, iu[0], iu[1]], tb[:, iu[0], iu[1]]
                worst = max(worst, (ta - tb).abs().max().item())
                t += 1
        errs.append(worst / (smax * smax if name == "sum (Im s)^2" else (1.0 if npl == 1 else smax)))
    print(f"C={C:4d} R={R:5d} n_obs={n_obs:6d} planes=0x{planes:x}: max |new - old| / max|S| per plane: " + "  ".join(f"{e:.2e}" for e in errs)
          + ("  NaN!" if not torch.isfinite(out).all() else ""))
    if timing:
        for name, fn in (("old (complex64)", lambda: engine.accumulate(sp, "trials_tapers", planes)),
                         ("new (planes)", lambda: _lib.check(lib.sc_fused2_csm_absim_f32(P.data_ptr(), byref(d), scale.data_ptr(), planes, out.data_ptr(), ws.data_ptr(), ws_bytes, None), "fused2"))):
            ts = []
            for rep in range(12):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            print(f"    {name}: {np.median(ts[2:]) * 1e3:.3f} ms")


if __name__ == "__main__":
    for C, R in ((128, 80), (64, 80), (96, 100), (32, 90), (100, 77), (128, 75), (128, 3), (64, 1), (20, 10), (130, 40), (160, 40),
                 (192, 30), (200, 33), (224, 20), (256, 40)):
        run(C, R)
    for C, R in ((128, 80), (60, 40), (160, 20), (256, 12)):
        run(C, R, planes=PLANES0 | _lib.PLANE_IM_SQ)
        run(C, R, planes=_lib.PLANE_SIGN_IM)
        run(C, R, planes=_lib.PLANE_CSM)
    run(128, 1000, F=129, W=7, timing=True)
    run(64, 2000, F=129, W=7, timing=True)
