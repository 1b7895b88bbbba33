"""Two (or more) ranks, each with its share of the trials, through parallel.sharded_measures(); rank 0 compares the
gathered coherence / wPLI with a single-process computation over all trials.  Run under torch.distributed.run;
SC_BENCH_BACKEND=gloo lets all ranks share one GPU (debug / CI on a 1-GPU box)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectral_connectivity_amd import _lib, engine, parallel  # noqa: E402
from spectral_connectivity_amd.transforms import dpss_windows  # noqa: E402


def main():
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    backend = os.environ.get("SC_BENCH_BACKEND", "nccl")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend != "nccl":
        local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    full_size = os.environ.get("SC_SHARD_FULL") == "1"      # the headline shape (cfg3): 1000 trials x 128 channels, 125 per rank at 8
    T, R, C, L, step, NW = (1024, 1000, 128, 256, 128, 4) if full_size else (512, 25, 64, 128, 64, 3)      # 25 trials over 2 ranks: unequal shards
    W = int(np.floor(T / step - L / step + 1))
    if full_size:
        from spectral_connectivity_amd.transforms import _make_tapers
        tap7 = _make_tapers(L, 1000.0, NW, 7)
        h = torch.from_numpy(np.ascontiguousarray(tap7.T / 1000.0, dtype=np.float32)).to(dev)
        g = torch.Generator(device=dev).manual_seed(1234)
        x_all = torch.randn((T, R, C), dtype=torch.float32, device=dev, generator=g)
        # the bench's synthetic input: white noise + a shared 60 Hz tone with the per-channel phase 2 pi c / C (a zero-lag
        # copy in every channel would make Im s a difference of large numbers at that bin: float32 rounding, in any device format,
        # then dominates the phase-lag measures -- not what this rehearsal is about)
        ph = 2 * np.pi * torch.arange(C, device=dev, dtype=torch.float32) / C
        x_all += 0.5 * torch.sin(2 * np.pi * 60.0 * torch.arange(T, device=dev, dtype=torch.float32)[:, None, None] / 1000.0 + ph[None, None, :])
        lo, hi = parallel.shard_bounds(R, world, rank)
        planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
        which = [_lib.M_COHERENCE_MAGNITUDE, _lib.M_WPLI]
        sp = engine.multitaper_spectra(x_all[:, lo:hi].contiguous(), h, L, step, L, W, "constant", planes_hint=planes)
        if rank != 0:
            del x_all
        got = parallel.sharded_measures(sp, planes, which, n_groups=4, equal_shards=(R % world == 0))
        torch.cuda.synchronize()
        if rank == 0:
            # against the FLOAT64 reference (tests/fp64_device_ref.py: torch.fft + einsum in float64, no product code), held to the
            # bound of tests/test_gpu_full_depth.py -- not against another run of the same kernels
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
            from fp64_device_ref import measures_fp64, spectra_fp64, sums_fp64
            X = spectra_fp64(x_all.cpu().numpy(), tap7, 1000.0, L, step, L)
            csm, ab = sums_fp64(X)
            n_obs = X.shape[2] * X.shape[3]
            del X
            assert n_obs == R * 7
            ref = measures_fp64(csm, ab, n_obs)
            worst = {}
            for gm, name in zip(got, ("coherence_magnitude", "weighted_phase_lag_index")):
                a, b = gm.cpu().numpy().astype(np.float64), ref[name]
                assert a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b))
                ok = ~np.isnan(b)
                scale = np.abs(b[ok]).max()
                worst[name] = (np.abs(a[ok] - b[ok]) / (3e-6 * np.abs(b[ok]) + 2e-7 * scale)).max()
                assert worst[name] <= 1.0, f"{name}: err / (3e-6 |ref| + 2e-7 max) = {worst[name]:.2f}"
            # and the single-process product path: the same spectra format, one record instead of eight summed blocks
            whole = engine.multitaper_spectra(x_all, h, L, step, L, W, "constant", planes_hint=planes)
            accum, n1 = engine.accumulate(whole, "trials_tapers", planes)
            for gm, w in zip(got, which):
                one = engine.measure(accum, C, planes, n1, w).reshape(W, L // 2 + 1, C, C)
                err = np.nanmax(np.abs(gm.cpu().numpy() - one.cpu().numpy()))
                assert err < 2e-5, f"measure {w}: {world} ranks vs one process, max err {err}"
            print(f"sharded_measures OK (full size, {world} ranks x {hi - lo} trials, exchange: {parallel.exchange_note()}); "
                  "err / (3e-6 |ref| + 2e-7 max|ref|) against the float64 reference: "
                  + ", ".join(f"{k} {v:.2f}" for k, v in worst.items()))
        dist.barrier()
        dist.destroy_process_group()
        return
    tap, _ = dpss_windows(L, NW, 5, is_low_bias=False)
    h = torch.from_numpy(np.ascontiguousarray(np.asarray(tap) * np.sqrt(200.0) / 200.0, dtype=np.float32)).to(dev)
    if h.shape[0] != 5:
        h = h.T.contiguous()
    x_all = torch.from_numpy(np.random.default_rng(0).standard_normal((T, R, C)).astype(np.float32)).to(dev)
    lo, hi = parallel.shard_bounds(R, world, rank)
    planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
    which = [_lib.M_COHERENCE_MAGNITUDE, _lib.M_WPLI]
    # the rank's own spectra in the planes format (f16 pieces, sc_fused2.hip) where it applies; the single-process reference
    # below goes through the complex64 kernels
    sp = engine.multitaper_spectra(x_all[:, lo:hi].contiguous(), h, L, step, L, W, "constant", planes_hint=planes)
    for groups in (1, 3, 4):
        got = parallel.sharded_measures(sp, planes, which, n_groups=groups)
        torch.cuda.synchronize()
        if rank == 0:
            full = engine.multitaper_spectra(x_all, h, L, step, L, W, "constant")
            accum, n_obs = engine.accumulate(full, "trials_tapers", planes)
            for g, w in zip(got, which):
                ref = engine.measure(accum, C, planes, n_obs, w).reshape(W, L // 2 + 1, C, C)
                a, b = g.cpu().numpy(), ref.cpu().numpy()
                assert a.shape == b.shape, (a.shape, b.shape)
                assert np.array_equal(np.isnan(a), np.isnan(b))
                err = np.nanmax(np.abs(a - b))
                assert err < 2e-5, f"groups={groups} measure {w}: max err {err}"
        else:
            assert all(g is None for g in got)
        dist.barrier()
    if rank == 0:
        print("sharded_measures OK")
    sharded_connectivity(world, rank, dev)
    dist.destroy_process_group()


def sharded_connectivity(world, rank, dev):
    """parallel.ShardedConnectivity: every rank builds it from ITS trials and calls the measures collectively; every
    rank must get what a single process gets from all the trials -- expectation measures (reduce-scatter, epilogue on
    the owned bins, all-gather), Granger (pairs dealt out over the ranks), canonical coherence (bins split)."""
    import spectral_connectivity_amd as sc
    rng = np.random.default_rng(11)
    T, R, C = 512, 13, 12                                     # 13 trials: unequal shards
    e = rng.standard_normal((T, R, C))
    x = np.zeros_like(e)
    for t in range(2, T):
        x[t] = 0.4 * x[t - 1] - 0.2 * x[t - 2] + e[t]
        x[t, :, 1:] += 0.3 * x[t - 1, :, :-1]
    kw = dict(sampling_frequency=200.0, time_halfbandwidth_product=2, n_time_samples_per_window=256)
    labels = np.repeat(np.arange(3), 4)
    lo, hi = parallel.shard_bounds(R, world, rank)
    for dtype, tol in ((np.complex64, 3e-5), (np.complex128, 1e-9)):
        mine = parallel.ShardedConnectivity.from_multitaper(sc.Multitaper(x[:, lo:hi], **kw), dtype=dtype)
        whole = sc.Connectivity.from_multitaper(sc.Multitaper(x, **kw), dtype=dtype)
        assert mine.n_observations == whole.n_observations
        for name in ("power", "coherency", "weighted_phase_lag_index", "phase_locking_value", "phase_lag_index"):
            a, b = getattr(mine, name)(), getattr(whole, name)()
            assert a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)), name
            ok = ~np.isnan(b)
            err = np.abs(a[ok] - b[ok]).max() / np.abs(b[ok]).max()
            bound = 4.0 / whole.n_observations if name == "phase_lag_index" else tol
            assert err <= bound, f"{name} ({np.dtype(dtype)}): {err}"
        a, b = mine.pairwise_spectral_granger_prediction(), whole.pairwise_spectral_granger_prediction()
        both = ~np.isnan(a) & ~np.isnan(b)
        flip = np.isnan(a) != np.isnan(b)          # gp <= 0 -> NaN: only values within the tolerance of 0 may flip
        assert np.all(np.abs(np.where(np.isnan(a), b, a)[flip]) <= 10 * tol * np.nanmax(b))
        assert np.abs(a[both] - b[both]).max() <= 10 * tol * np.nanmax(b)
        a, la = mine.canonical_coherence(labels)
        b, lb = whole.canonical_coherence(labels)
        ok = ~np.isnan(b)
        assert np.array_equal(la, lb) and np.array_equal(np.isnan(a), np.isnan(b))
        assert np.abs(a[ok] - b[ok]).max() <= 10 * tol
    # n_observations per expectation type (round-2 advisor finding): sliding windows, "time_trials_tapers", canonical
    # coherence (always trials x tapers) BEFORE power (windows x trials x tapers) -- each must use its own count
    kw2 = dict(sampling_frequency=200.0, time_halfbandwidth_product=2, n_time_samples_per_window=128,
               n_time_samples_per_step=128)
    mine = parallel.ShardedConnectivity.from_multitaper(sc.Multitaper(x[:, lo:hi], **kw2),
                                                        expectation_type="time_trials_tapers")
    whole = sc.Connectivity.from_multitaper(sc.Multitaper(x, **kw2), expectation_type="time_trials_tapers")
    a, _ = mine.canonical_coherence(labels)
    b, _ = whole.canonical_coherence(labels)
    ok = ~np.isnan(b)
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.abs(a[ok] - b[ok]).max() <= 1e-8
    a, b = mine.power(), whole.power()
    assert mine.n_observations == whole.n_observations and np.abs(a - b).max() <= 1e-9 * np.abs(b).max()
    dist.barrier()
    if rank == 0:
        print("ShardedConnectivity OK")


if __name__ == "__main__":
    main()
