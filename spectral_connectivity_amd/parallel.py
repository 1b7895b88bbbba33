"""Trial sharding over the GPUs of one node (one process per GPU, torch.distributed / RCCL).

The expectation over trials x tapers is a plain sum of per-observation terms for every
accumulator plane (reference connectivity.py:67-75, :489), so trials shard embarrassingly:
each rank runs stages A and B on its own trials, then ONE exchange sums the un-normalised
accumulator records.  The exchange is a reduce-scatter over frequency/window bins (each
rank ends up owning 1/N of the bins, summed over all ranks), the measures epilogue runs on
the owned bins only, and a gather assembles the final measures on the rank that owns the
user-facing result (or an all-gather when every rank needs them) -- (N-1)/N of the records
per link instead of the 2x of an all-reduce followed by a redundant epilogue.  Division by
n_observations happens after the sum (never average ratios).
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items, world_size, rank):
    """Contiguous balanced partition [lo, hi) of n_items (first n_items % world get one more)."""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def padded_bins(n_bins, world_size):
    return (n_bins + world_size - 1) // world_size * world_size


def reduce_scatter_bins(accum, group=None):
    """Sum accumulator records over ranks; return (this rank's bin shard, bin_lo, bin_hi).

    ``accum``: [n_bins, floats_per_bin] float32.  Bins are padded to a multiple of the
    world size so every rank owns the same count; ``bin_hi`` is clipped to n_bins.
    """
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return accum, 0, accum.shape[0]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n_bins, fpb = accum.shape
    per = padded_bins(n_bins, world) // world
    if per * world != n_bins:
        pad = torch.zeros((per * world - n_bins, fpb), dtype=accum.dtype, device=accum.device)
        accum = torch.cat([accum, pad], dim=0)
    lo = rank * per
    hi = min(lo + per, n_bins)
    if dist.get_backend(group) == "gloo":      # CPU tests / debug runs: gloo has no reduce_scatter
        if accum.is_cuda:                      # gloo moves CUDA tensors through the host
            host = accum.cpu()
            dist.all_reduce(host, group=group)
            shard = host[lo:lo + per].to(accum.device)
        else:
            dist.all_reduce(accum, group=group)
            shard = accum[lo:lo + per].clone()
    else:
        shard = torch.empty((per, fpb), dtype=accum.dtype, device=accum.device)
        dist.reduce_scatter_tensor(shard, accum, group=group)
    return shard, lo, max(hi, lo)


def all_gather_bins(shard_out, n_bins, group=None):
    """Assemble per-rank measure shards [per, ...] into the full [n_bins, ...] tensor."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return shard_out[:n_bins]
    world = dist.get_world_size(group)
    full = torch.empty((shard_out.shape[0] * world,) + tuple(shard_out.shape[1:]),
                       dtype=shard_out.dtype, device=shard_out.device)
    dist.all_gather_into_tensor(full, shard_out.contiguous(), group=group)
    return full[:n_bins]


def gather_bins(shard_out, n_bins, dst=0, group=None):
    """Assemble the per-rank measure shards on ONE rank (the process that owns the user-facing result).

    Over xGMI every rank sends its 1/N directly to `dst` on its own link (N-1 concurrent transfers of
    1/N of the data) instead of the N-1 ring steps of an all-gather.  Returns the [n_bins, ...] tensor
    on `dst`, None elsewhere.
    """
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return shard_out[:n_bins]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = shard_out.device
    shard_out = shard_out.contiguous()
    if dist.get_backend(group) == "gloo" and shard_out.is_cuda:
        shard_out = shard_out.cpu()
    parts = [torch.empty_like(shard_out) for _ in range(world)] if rank == dst else None
    dist.gather(shard_out, parts, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat(parts, dim=0)[:n_bins].to(dev)


def total_observations(local_n_obs, group=None):
    """n_observations of the whole job = sum of the shards' counts (trials differ per rank)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return int(local_n_obs)
    t = torch.tensor([int(local_n_obs)], dtype=torch.int64)
    if dist.get_backend(group) != "gloo":
        t = t.cuda()
    dist.all_reduce(t, group=group)
    return int(t.item())


_side_streams = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


def sharded_measures(spectra, planes, which, n_groups=4, dst=0, group=None, mark=None, equal_shards=False):
    """Stage B + exchange + epilogue for trial-sharded spectra, pipelined over frequency groups.

    Every rank holds the spectra of ITS trials.  The frequency axis is cut into ``n_groups`` ranges; for
    each range the rank accumulates its un-normalised records on the launch stream, and a second stream
    takes the range through reduce-scatter (sum over ranks, 1/N of the bins each) -> measures epilogue on
    the owned bins -> gather on ``dst`` while the launch stream is already accumulating the next range:
    only the last range's exchange is exposed.  ``which``: list of ``_lib.M_*`` measures.  Returns, on
    ``dst``, one tensor per measure shaped [n_windows, n_freq, C, C] (None on the other ranks).
    ``equal_shards``: every rank holds the same number of trials, so n_observations = local count x world
    without a collective (otherwise one small all-reduce per call).
    """
    from . import engine
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    F, W, C = spectra.F, spectra.W, spectra.C
    n_groups = max(1, min(int(n_groups), F))
    main = torch.cuda.current_stream()
    side = _side_stream(spectra.X.device) if world > 1 else main
    parts = [[] for _ in which]
    n_local = spectra.R * spectra.K
    n_total = total_observations_equal(n_local, world) if equal_shards else total_observations(n_local, group)
    for g in range(n_groups):
        f0, f1 = shard_bounds(F, n_groups, g)
        accum, n_obs = engine.accumulate(spectra.freq_slice(f0, f1), "trials_tapers", planes, mark=mark)
        n_bins = accum.shape[0]
        if world > 1:
            ready = torch.cuda.Event()
            ready.record(main)
            accum.record_stream(side)
        with torch.cuda.stream(side):
            if world > 1:
                side.wait_event(ready)
            shard, lo, hi = reduce_scatter_bins(accum, group)
            for m, w in enumerate(which):
                out = engine.measure(shard, C, planes, n_total, w)
                if world > 1:
                    out = gather_bins(out, n_bins, dst=dst, group=group)
                if out is not None:
                    parts[m].append(out.reshape(W, f1 - f0, *out.shape[1:]))
    if world > 1:
        main.wait_stream(side)
    if mark:
        mark("exchange_epilogue_tail")
    if rank != dst:
        return [None for _ in which]
    result = []
    for chunks in parts:
        if world > 1:
            for c in chunks:                  # produced on the side stream, consumed on the launch stream
                c.record_stream(main)
        result.append(chunks[0] if len(chunks) == 1 else torch.cat(chunks, dim=1))
    return result


def total_observations_equal(local_n_obs, world):
    """n_observations of the job when every rank holds the same number of trials (no collective needed)."""
    return int(local_n_obs) * int(world)
