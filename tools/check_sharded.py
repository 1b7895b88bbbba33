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
    T, R, C, L, step, NW = 512, 25, 64, 128, 64, 3      # 25 trials over 2 ranks: unequal shards
    W = int(np.floor(T / step - L / step + 1))
    tap, _ = dpss_windows(L, NW, 5, is_low_bias=False)
    h = torch.from_numpy(np.ascontiguousarray(np.asarray(tap) * np.sqrt(200.0) / 200.0, dtype=np.float32)).to(dev)
    if h.shape[0] != 5:
        h = h.T.contiguous()
    x_all = torch.from_numpy(np.random.default_rng(0).standard_normal((T, R, C)).astype(np.float32)).to(dev)
    lo, hi = parallel.shard_bounds(R, world, rank)
    planes = _lib.PLANE_CSM | _lib.PLANE_ABS_IM
    which = [_lib.M_COHERENCE_MAGNITUDE, _lib.M_WPLI]
    sp = engine.multitaper_spectra(x_all[:, lo:hi].contiguous(), h, L, step, L, W, "constant")
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
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
