"""A/B: stage A of window w + 1 beside stage B of window w (cfg3: 7 windows), on two streams with compute-unit masks
(hipExtStreamCreateWithCUMask).  Stage A is bound by its store stream and leaves the matrix pipes idle; stage B is issue-bound at
2.1 TB/s -- but each takes a whole compute unit's LDS (2 x 75 KB and 153 KB), so they can only run side by side on DISJOINT compute
units.  Prints: each stage alone per window under a mask of n compute units; the sequential chain on one stream; the pipelined chain
for several splits.  Results of every variant are checked against the one-launch chain.
    python tools/overlap_ab.py"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine      # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
lib = _lib.load()
PL = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
T, R, C, L, step, K = 1024, 1000, 128, 256, 128, 7
N, W, F = L, 7, 129
torch.manual_seed(0)
x = torch.randn((T, R, C), device=dev)
t = torch.arange(T, device=dev) / 1000.0
x += 0.5 * torch.sin(2 * np.pi * 60.0 * t[:, None, None] + 2 * np.pi * torch.arange(C, device=dev)[None, None, :] / C)
from spectral_connectivity_amd.transforms import _make_tapers      # noqa: E402
tapers = _make_tapers(L, 1000.0, 4.0, K)
h = torch.from_numpy(np.ascontiguousarray(tapers.T / 1000.0, dtype=np.float32)).to(dev)
tw = engine.twiddles(N, dev)
row_bytes = int(lib.sc_planes_row_bytes(C))


def masked_stream(cus):
    """A stream restricted to the compute units `cus` (indices into the 256-bit mask)."""
    words = (ctypes.c_uint32 * 8)()
    for c in cus:
        words[c // 32] |= 1 << (c % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def scales():
    scale = torch.empty((2 * C,), dtype=torch.float32, device=dev)
    wb = int(lib.sc_planes_scales_work_bytes(T * R, C))
    work = torch.empty((wb,), dtype=torch.uint8, device=dev)
    q = torch.empty((1,), dtype=torch.float32, device=dev)
    abs_sum, _ = engine._taper_norms(h)
    _lib.check(lib.sc_planes_scales_quality_f32(x.data_ptr(), T, R, C, _lib.DETREND["constant"], abs_sum, scale.data_ptr(), work.data_ptr(),
                                                wb, q.data_ptr(), torch.cuda.current_stream().cuda_stream), "scales")
    return scale


def stage_a(w, P, scale, stream):
    xw = x[w * step:]
    _lib.check(lib.sc_multitaper_fft_planes_f32(xw.data_ptr(), L, R, C, L, step, 1, N, h.data_ptr(), K, _lib.DETREND["constant"],
                                                tw.data_ptr(), scale.data_ptr(), P.data_ptr(), stream.cuda_stream), "stage A")


def stage_b(P, scale, stream):
    sp = engine.DeviceSpectra(None, (F, 1, R, K, C), (R * K * C, R * K * C, K * C, C), N, True, C_alloc=C, P=P, scale=scale)
    with torch.cuda.stream(stream):
        accum, n_obs = engine.accumulate(sp, "trials_tapers", PL, fold=False)
        return engine.measure_multi(accum, C, PL, n_obs, [_lib.M_COHERENCE_MAGNITUDE, _lib.M_WPLI])


def chain(sa, sb, pipelined):
    """All seven windows; returns the per-window measures."""
    cur = torch.cuda.current_stream()
    scale = scales()
    ev0 = torch.cuda.Event(); ev0.record(cur)
    sa.wait_event(ev0); sb.wait_event(ev0)
    Ps = [torch.empty((F * R * K * row_bytes,), dtype=torch.uint8, device=dev) for _ in range(2 if pipelined else 1)]
    outs, done_b = [], [None, None]
    for w in range(W):
        P = Ps[w % len(Ps)]
        if done_b[w % len(Ps)] is not None:
            sa.wait_event(done_b[w % len(Ps)])          # the buffer's previous window has been consumed
        stage_a(w, P, scale, sa)
        e = torch.cuda.Event(); e.record(sa)
        sb.wait_event(e)
        outs.append(stage_b(P, scale, sb))
        d = torch.cuda.Event(); d.record(sb)
        done_b[w % len(Ps)] = d
        for o in Ps:
            o.record_stream(sa); o.record_stream(sb)
    cur.wait_stream(sa); cur.wait_stream(sb)
    return outs


def timed(f, reps=6):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


def one_launch():
    sp = engine.multitaper_spectra(x, h, L, step, N, W, "constant", planes_hint=PL)
    accum, n_obs = engine.accumulate(sp, "trials_tapers", PL, fold=False)
    del sp
    return engine.measure_multi(accum, C, PL, n_obs, [_lib.M_COHERENCE_MAGNITUDE, _lib.M_WPLI])


ref = one_launch()
print(f"one launch per stage (the bench chain): {timed(one_launch):.2f} ms")
full = masked_stream(range(256))
def same(outs):
    """Largest difference from the one-launch chain over both measures (a window on its own splits its bins over other workgroup
    counts: another summation order, float32 rounding apart)."""
    worst = 0.0
    for k in range(2):
        r = ref[k].reshape(W, F, C, C)
        for w in range(W):
            worst = max(worst, (outs[w][k].reshape(F, C, C).nan_to_num() - r[w].nan_to_num()).abs().max().item())
    return worst


outs = chain(full, full, False)
print(f"per-window launches, one stream, all 256 CUs: {timed(lambda: chain(full, full, False)):.2f} ms   (max |difference| from the one-launch chain: {same(outs):.1e})")

# which compute units a mask bit names is not documented for this part: two layouts are tried -- the first n bits, and n / 8 bits
# out of every 32 (one slice of every XCD if the bits go XCD by XCD)
def first(n):
    return list(range(n))


def spread(n):
    per = n // 8
    return [32 * g + i for g in range(8) for i in range(per)]


def rest(cus):
    s = set(cus)
    return [c for c in range(256) if c not in s]


scale0 = scales()
P0 = torch.empty((F * R * K * row_bytes,), dtype=torch.uint8, device=dev)
print("# one window's stage A / stage B alone under a mask of n compute units (ms); layout 'first' / 'spread'")
for n in (256, 192, 160, 128, 96, 64, 32):
    row = []
    for lay in (first, spread):
        s = masked_stream(lay(n))
        ta = timed(lambda: stage_a(3, P0, scale0, s))
        tb = timed(lambda: stage_b(P0, scale0, s))
        row.append(f"{lay.__name__}: A {ta:5.3f} B {tb:5.3f}")
    print(f"  n={n:3d}  " + "   ".join(row), flush=True)

print("# pipelined: stage A of window w + 1 on `a` compute units beside stage B of window w on the other 256 - a (two streams); ms per step")
for lay in (first, spread):
    for a in (32, 64, 96, 128):
        sa, sb = masked_stream(lay(a)), masked_stream(rest(lay(a)))
        outs = chain(sa, sb, True)
        torch.cuda.synchronize()
        print(f"  {lay.__name__:6s} A on {a:3d} / B on {256 - a:3d}: {timed(lambda: chain(sa, sb, True)):.2f} ms   (max |difference|: {same(outs):.1e})", flush=True)
    sa = sb2 = None
# no masks, two streams: whatever the dispatcher does with two kernels that each want every compute unit's LDS
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
print(f"  two unmasked streams: {timed(lambda: chain(s1, s2, True)):.2f} ms")
