"""N>1 path on CPU: two gloo ranks shard the trials, accumulate UN-normalised records, run
parallel.reduce_scatter_bins / all_gather_bins, and must reproduce the single-process result.
The HIP kernels cannot run here, so each rank's stage-B output is produced by the NumPy
oracle (as the checker's stand-in for the kernel, in the packed tile layout of sc_hip.h);
what is under test is the sharding, padding, reduction, bin ownership and gathering logic
that bench.py and the multi-GPU path use verbatim."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pack_tiles(csm):
    """(n_bins, C, C) complex sums -> [n_bins, 2 planes * n_tiles * 256] float32 records."""
    n_bins, C, _ = csm.shape
    NB = (C + 15) // 16
    pad = np.zeros((n_bins, NB * 16, NB * 16), dtype=complex)
    pad[:, :C, :C] = csm
    tiles = [pad[:, bi * 16:(bi + 1) * 16, bj * 16:(bj + 1) * 16] for bi in range(NB) for bj in range(bi, NB)]
    t = np.stack(tiles, axis=1).reshape(n_bins, len(tiles) * 256)
    return np.concatenate([t.real, t.imag], axis=1).astype(np.float32)


def unpack_diag_power(rec, C):
    NB = (C + 15) // 16
    n_tiles = NB * (NB + 1) // 2
    re = rec[:, : n_tiles * 256].reshape(rec.shape[0], n_tiles, 16, 16)
    out = np.zeros((rec.shape[0], C))
    for c in range(C):
        b = c // 16
        t = b * NB - b * (b - 1) // 2
        out[:, c] = re[:, t, c % 16, c % 16]
    return out


def worker(rank, world, port, x, ref_rec, ref_power, errors):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle import spectral_oracle as so
        from spectral_connectivity_amd import parallel
        R, C = x.shape[1], x.shape[2]
        lo, hi = parallel.shard_bounds(R, world, rank)
        coef, _ = so.multitaper_fft(x[:, lo:hi], fs=100.0, NW=2, n_time_samples_per_window=32)
        F = coef.shape[3] // 2 + 1
        n_local = coef.shape[1] * coef.shape[2]
        # un-normalised sums over this rank's observations, bins = window*F + f
        sums = np.einsum("wrkni,wrknj->wnij", coef[..., :F, :], coef[..., :F, :].conj()).reshape(-1, C, C)
        accum = torch.from_numpy(pack_tiles(sums))
        n_bins = accum.shape[0]
        shard, b_lo, b_hi = parallel.reduce_scatter_bins(accum)
        n_total = parallel.total_observations(n_local)
        assert n_total == R * coef.shape[2]
        per = parallel.padded_bins(n_bins, world) // world
        assert shard.shape[0] == per and b_lo == rank * per
        np.testing.assert_allclose(shard[: b_hi - b_lo].numpy(), ref_rec[b_lo:b_hi], rtol=2e-5, atol=1e-4)
        # "epilogue" on the owned bins (power = diagonal / n), then all-gather
        power = torch.zeros((per, C), dtype=torch.float64)
        power[: b_hi - b_lo] = torch.from_numpy(unpack_diag_power(shard[: b_hi - b_lo].numpy(), C) / n_total)
        full = parallel.all_gather_bins(power, n_bins)
        np.testing.assert_allclose(full.numpy(), ref_power, rtol=2e-5)
        on0 = parallel.gather_bins(power, n_bins, dst=0)
        if rank == 0:
            np.testing.assert_allclose(on0.numpy(), ref_power, rtol=2e-5)
        else:
            assert on0 is None
        dist.destroy_process_group()
    except Exception as exc:  # surface the failure in the parent
        errors.put(f"rank {rank}: {exc!r}")
        raise


@pytest.mark.parametrize("world,R", [(2, 6), (2, 5)])
def test_trial_sharded_reduce_scatter_matches_single_process(world, R):
    sys.path.insert(0, ROOT)
    from oracle import spectral_oracle as so
    rng = np.random.default_rng(7)
    C = 20
    x = rng.standard_normal((96, R, C))
    coef, _ = so.multitaper_fft(x, fs=100.0, NW=2, n_time_samples_per_window=32)
    F = coef.shape[3] // 2 + 1
    sums = np.einsum("wrkni,wrknj->wnij", coef[..., :F, :], coef[..., :F, :].conj()).reshape(-1, C, C)
    ref_rec = pack_tiles(sums)
    ref_power = so.power(coef).reshape(-1, C)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    errors = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, x, ref_rec, ref_power, errors)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    msgs = []
    while not errors.empty():
        msgs.append(errors.get())
    assert not msgs, msgs
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]


def test_shard_bounds_cover_everything():
    from spectral_connectivity_amd import parallel
    for n in (1, 7, 8, 1000):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
