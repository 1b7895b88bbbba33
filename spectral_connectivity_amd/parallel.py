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
import os

import torch
import torch.distributed as dist


def _no_exchange(group=None):
    """True when there is nothing to exchange: no process group, or a single rank.  SC_FORCE_EXCHANGE=1 sends a
    one-rank group through the collectives anyway -- the rehearsal of the RCCL calls, streams and buffers on a box
    with one GPU (tests/test_gpu_configs.py::test_rccl_exchange_path_world_one)."""
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size(group) == 1 and os.environ.get("SC_FORCE_EXCHANGE", "0") != "1"


def shard_bounds(n_items, world_size, rank):
    """Contiguous balanced partition [lo, hi) of n_items (first n_items % world get one more)."""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def padded_bins(n_bins, world_size):
    return (n_bins + world_size - 1) // world_size * world_size


def exchange_algorithm():
    """How the records are summed over the ranks (``SC_EXCHANGE``):

    ``direct`` (default): every rank sends bin block j of its record straight to rank j -- ``all_to_all_single``, N-1
    concurrent point-to-point transfers of 1/N of the record, one per xGMI link -- and adds the N blocks it received in
    rank order (the same sum on every run).  xGMI is a full mesh of point-to-point links: the N-1 steps of a ring
    reduce-scatter each move 1/N of the record over ONE link while the other six idle, the direct form moves the same
    bytes over all of them at once (100 MB of records at cfg3 on 8 GPUs: 12.5 MB per link once, instead of seven times
    in a row).
    ``ring``: ``reduce_scatter_tensor``, whatever algorithm RCCL picks (gloo: all-reduce + slice).
    ``library``: the direct exchange through the C ABI's own RCCL communicator (``sc_comm_exchange_blocks_f32``, sc_comm.hip:
    grouped ncclSend / ncclRecv) instead of ``torch.distributed`` -- what a host without torch would call; torch only carries
    the 128-byte communicator id to the ranks once."""
    mode = os.environ.get("SC_EXCHANGE", "direct")
    if mode not in ("direct", "ring", "library"):
        raise ValueError(f"SC_EXCHANGE={mode!r}: expected 'direct', 'ring' or 'library'")
    return mode                    # (a group whose ranks AGREED that the direct form fails takes the ring: _ring_agreed, per group)


_direct_failed = []                # [message] of every failure of the direct exchange in this process (exchange_note reports the first)
_ring_agreed = set()               # (process group, algorithm) keys whose ranks have agreed to take the library reduce-scatter instead


def exchange_note():
    """What the bench line / logs should say about the exchange that really ran."""
    if _ring_agreed:
        return "ring: reduce_scatter_tensor (the direct all_to_all_single exchange raised: " + (_direct_failed[0] if _direct_failed else "?") + ")"
    if exchange_algorithm() == "library":
        return ("library: sc_comm_exchange_blocks_f32 (grouped ncclSend / ncclRecv of the 1/N bin blocks through the C ABI's own "
                "communicator), summed in rank order inside the epilogue kernel")
    return ("direct: all_to_all_single of the 1/N bin blocks (one xGMI link each), summed in rank order inside the epilogue kernel"
            if exchange_algorithm() == "direct" else "ring: reduce_scatter_tensor")


XGMI_LINK_GBS = 64.0               # per direction and link, sustained (153 GB/s bidirectional peak per link: ~40 % of it one way)


def exchange_model(record_bytes, measure_bytes, world):
    """Bytes per xGMI link and the time they take at XGMI_LINK_GBS for one step's exchange: the direct reduce-scatter sends 1/N of
    the record to each peer (N - 1 links at once), a ring moves (N - 1)/N of it over one link in N - 1 steps; the gather sends
    each rank's 1/N of the measures to rank 0 (its N - 1 incoming links at once)."""
    if world <= 1:
        return None
    direct = record_bytes / world
    ring = record_bytes * (world - 1) / world
    gather = measure_bytes / world
    return {"record_bytes": int(record_bytes), "bytes_per_link_direct": int(direct), "bytes_per_link_ring_total": int(ring),
            "gather_bytes_per_link": int(gather), "link_gb_per_s_assumed": XGMI_LINK_GBS,
            "predicted_ms_direct": round((direct + gather) / (XGMI_LINK_GBS * 1e9) * 1e3, 4),
            "predicted_ms_ring": round((ring + gather) / (XGMI_LINK_GBS * 1e9) * 1e3, 4)}


_library_comms = {}                # process group -> sc_comm handle of this process (created once, on the current device)


def _library_comm(group, device):
    """The C ABI's RCCL communicator over the ranks of ``group``: rank 0 draws the id (sc_comm_unique_id), torch.distributed
    carries its 128 bytes to the others, every rank joins on its current device (sc_comm_create)."""
    import ctypes
    from . import _lib
    key = id(group) if group is not None else 0
    comm = _library_comms.get(key)
    if comm is None:
        lib = _lib.load()
        if not lib.sc_comm_available():
            raise RuntimeError("SC_EXCHANGE=library: RCCL (librccl.so.1) could not be loaded by libsc_hip.so")
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        buf = ctypes.create_string_buffer(128)
        if rank == 0:
            _lib.check(lib.sc_comm_unique_id(buf), "sc_comm_unique_id")
        box = [bytes(buf.raw)]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.sc_comm_create(ctypes.create_string_buffer(box[0], 128), world, rank, ctypes.byref(handle)), "sc_comm_create")
        comm = _library_comms[key] = handle
    return comm


def _library_blocks(accum, world, per, fpb, group):
    """The direct exchange through sc_comm_exchange_blocks_f32 (records on the GPU, float32); None when it cannot run."""
    from . import _lib
    if not accum.is_cuda or accum.dtype != torch.float32:
        _direct_failed.append("SC_EXCHANGE=library needs float32 records on the GPU")
        return None
    try:
        comm = _library_comm(group, accum.device)
        src = accum.contiguous()
        recv = torch.empty_like(src)
        _lib.check(_lib.load().sc_comm_exchange_blocks_f32(comm, src.data_ptr(), recv.data_ptr(), per * fpb,
                                                           torch.cuda.current_stream(accum.device).cuda_stream), "sc_comm_exchange_blocks_f32")
    except (RuntimeError, _lib.HipEngineError) as exc:
        _direct_failed.append(str(exc).splitlines()[0][:200])
        return None
    recv.record_stream(torch.cuda.current_stream(accum.device))
    return recv.view(world, per, fpb)


def _direct_blocks(accum, world, per, fpb, group):
    """all_to_all_single of the bin blocks; [world, per, fpb] on the records' device, or None when the backend refuses (RCCL
    builds without all-to-all, ...): the caller then takes the library reduce-scatter, and says so (exchange_note)."""
    via_host = dist.get_backend(group) == "gloo" and accum.is_cuda          # gloo moves CUDA tensors through the host
    src = (accum.cpu() if via_host else accum).contiguous()
    recv = torch.empty_like(src)
    try:
        dist.all_to_all_single(recv, src, group=group)                      # block j of every rank's record -> rank j
    except RuntimeError as exc:
        _direct_failed.append(str(exc).splitlines()[0][:200])
        return None
    blocks = recv.view(world, per, fpb)
    return blocks.to(accum.device) if via_host else blocks


_direct_agreed = set()              # process groups whose ranks have agreed that the direct exchange works


def _agreed_direct_exchange(accum, world, per, fpb, group):
    """The direct exchange, with the choice of algorithm AGREED over the ranks: the first exchange of a process group is followed
    by one all-reduce of a failure flag -- if the direct form raised on ANY rank (a backend without all-to-all, RCCL missing for
    SC_EXCHANGE=library on one of them) every rank takes the library reduce-scatter from then on, so that no two ranks ever issue
    different collectives.  Once agreed, a failure of the direct exchange is an error (raised), not a silent switch."""
    # (the C ABI's exchange carries float32 records on the GPU; double records of the float64 engine, or records on the host,
    #  take the torch form -- the same on every rank, so not a failure)
    library = exchange_algorithm() == "library" and accum.is_cuda and accum.dtype == torch.float32
    fn = _library_blocks if library else _direct_blocks
    key = (id(group) if group is not None else 0, library)
    if key in _ring_agreed:            # decided for this group, on every rank at the same exchange: no collective here
        return None
    if key in _direct_agreed:
        blocks = fn(accum, world, per, fpb, group)
        if blocks is None:
            raise RuntimeError("the direct exchange failed after the ranks had agreed on it: " + (_direct_failed[-1] if _direct_failed else "?"))
        return blocks
    blocks = fn(accum, world, per, fpb, group)
    flag = torch.tensor([0 if blocks is not None else 1], dtype=torch.int32)
    if dist.get_backend(group) != "gloo":
        flag = flag.to(accum.device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    if int(flag.item()):
        if blocks is not None or not _direct_failed:
            _direct_failed.append("the direct exchange raised on another rank")
        _ring_agreed.add(key)          # every rank of the group passes through here at this exchange: the same decision everywhere
        return None
    _direct_agreed.add(key)
    return blocks


def reduce_scatter_bins(accum, group=None, keep_parts=False):
    """Sum accumulator records over ranks; return (this rank's bin shard, bin_lo, bin_hi).

    ``accum``: [n_bins, floats_per_bin] float32.  Bins are padded to a multiple of the
    world size so every rank owns the same count; ``bin_hi`` is clipped to n_bins.
    ``keep_parts`` (direct exchange only): return the N received blocks [world, per, fpb] unsummed -- the epilogue
    kernel adds them in rank order while it reads them (engine.measure_multi), one pass and one buffer less.
    """
    if _no_exchange(group):
        return accum, 0, accum.shape[0]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n_bins, fpb = accum.shape
    per = padded_bins(n_bins, world) // world
    if per * world != n_bins:
        base = accum._base
        if (base is not None and base.dim() == 2 and base.shape == (per * world, fpb) and base.is_contiguous()
                and base.data_ptr() == accum.data_ptr()):
            accum = base                       # engine.accumulate(row_multiple=world): the zeroed tail rows are already there
        else:
            pad = torch.zeros((per * world - n_bins, fpb), dtype=accum.dtype, device=accum.device)
            accum = torch.cat([accum, pad], dim=0)
    lo = rank * per
    hi = min(lo + per, n_bins)
    if exchange_algorithm() in ("direct", "library"):
        blocks = _agreed_direct_exchange(accum, world, per, fpb, group)
        if blocks is not None:
            if keep_parts:
                return blocks, lo, max(hi, lo)                              # [world, per, fpb]: summed by the consumer, in rank order
            shard = blocks[0].clone()                                       # (a fresh [per, fpb]: the receive buffer can go)
            for k in range(1, world):                                       # rank order
                shard.add_(blocks[k])
            return shard, lo, max(hi, lo)
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
    if _no_exchange(group):
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
    if _no_exchange(group):
        return shard_out[:n_bins]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = shard_out.device
    shard_out = shard_out.contiguous()
    if dist.get_backend(group) == "gloo" and shard_out.is_cuda:
        shard_out = shard_out.cpu()
    # the shards land in consecutive slices of ONE buffer on dst: no concatenation afterwards
    full = torch.empty((world * shard_out.shape[0],) + tuple(shard_out.shape[1:]), dtype=shard_out.dtype,
                       device=shard_out.device) if rank == dst else None
    parts = list(full.split(shard_out.shape[0], dim=0)) if rank == dst else None
    dist.gather(shard_out, parts, dst=dst, group=group)
    if rank != dst:
        return None
    return full[:n_bins].to(dev)


def total_observations(local_n_obs, group=None):
    """n_observations of the whole job = sum of the shards' counts (trials differ per rank)."""
    if _no_exchange(group):
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


def sharded_measures(spectra, planes, which, n_groups=4, dst=0, group=None, mark=None, equal_shards=False,
                     timing=None):
    """Stage B + exchange + epilogue for trial-sharded spectra, pipelined over frequency groups.

    Every rank holds the spectra of ITS trials.  The frequency axis is cut into ``n_groups`` ranges; for
    each range the rank accumulates its un-normalised records on the launch stream, and a second stream
    takes the range through reduce-scatter (sum over ranks, 1/N of the bins each) -> measures epilogue on
    the owned bins -> gather on ``dst`` while the launch stream is already accumulating the next range:
    only the last range's exchange is exposed.  ``which``: list of ``_lib.M_*`` measures.  Returns, on
    ``dst``, one tensor per measure shaped [n_windows, n_freq, C, C] (None on the other ranks).
    ``equal_shards``: every rank holds the same number of trials, so n_observations = local count x world
    without a collective (otherwise one small all-reduce per call).
    ``timing``: a list; one dict per call is appended with ``collective_ms`` (time inside reduce-scatter and gather on
    the exchange stream), ``exposed_ms`` (what the launch stream waited for after its last accumulation: the part of
    exchange + epilogue that did not hide behind stage B), ``bytes_reduced`` and ``n_groups`` -- read after the next
    device synchronisation (the values are filled in lazily from events).
    """
    from . import engine
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    xchg = not _no_exchange(group)             # world > 1, or a one-rank rehearsal of the exchange
    F, W, C = spectra.F, spectra.W, spectra.C
    n_groups = max(1, min(int(n_groups), F))
    main = torch.cuda.current_stream()
    side = _side_stream(spectra.device) if xchg else main
    # the measures of every frequency range land in ONE preallocated [W, F, ...] tensor per measure on `dst` (a strided
    # device copy per range), not in a list that is concatenated afterwards
    result = [None for _ in which]
    n_local = spectra.R * spectra.K
    n_total = total_observations_equal(n_local, world) if equal_shards else total_observations(n_local, group)
    coll_events, bytes_reduced = [], 0

    def ev(stream):
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)
        return e

    for g in range(n_groups):
        f0, f1 = shard_bounds(F, n_groups, g)
        accum, n_obs = engine.accumulate(spectra.freq_slice(f0, f1), "trials_tapers", planes, mark=mark, row_multiple=world)
        n_bins = accum.shape[0]
        if xchg:
            ready = torch.cuda.Event()
            ready.record(main)
            accum.record_stream(side)
        with torch.cuda.stream(side):
            if xchg:
                side.wait_event(ready)
            t0 = ev(side) if timing is not None else None
            shard, lo, hi = reduce_scatter_bins(accum, group, keep_parts=True)
            if timing is not None:
                coll_events.append((t0, ev(side)))
                bytes_reduced += accum.numel() * accum.element_size()
            outs = engine.measure_multi(shard, C, planes, n_total, which, stacked=xchg)   # one launch where the measures allow it
            block = outs[0]._base if xchg and len(outs) > 1 else None
            if block is not None and block.dim() == 4 and block.shape[0] == len(which) and all(
                    o._base is block for o in outs):
                # every measure of the range in ONE gather: the ranks' [n_measures, per, C, C] blocks land side by side on dst
                t0 = ev(side) if timing is not None else None
                full = gather_bins(block, block.shape[0] * world, dst=dst, group=group)     # [world * n_measures, per, C, C]
                if timing is not None:
                    coll_events.append((t0, ev(side)))
                if full is not None:
                    per = block.shape[1]
                    full = full.reshape(world, len(which), per, *block.shape[2:])
                    for m in range(len(which)):
                        out = full[:, m].reshape(world * per, *block.shape[2:])[:n_bins]
                        if result[m] is None:
                            result[m] = torch.empty((W, F) + tuple(out.shape[1:]), dtype=out.dtype, device=out.device)
                            result[m].record_stream(main)       # filled on the exchange stream, used on the launch stream
                        result[m][:, f0:f1].copy_(out.reshape(W, f1 - f0, *out.shape[1:]))
                continue
            for m, w in enumerate(which):
                out = outs[m]
                if xchg:
                    t0 = ev(side) if timing is not None else None
                    out = gather_bins(out, n_bins, dst=dst, group=group)
                    if timing is not None:
                        coll_events.append((t0, ev(side)))
                if out is not None:
                    if result[m] is None:
                        result[m] = torch.empty((W, F) + tuple(out.shape[1:]), dtype=out.dtype, device=out.device)
                        if xchg:
                            result[m].record_stream(main)       # filled on the exchange stream, used on the launch stream
                    result[m][:, f0:f1].copy_(out.reshape(W, f1 - f0, *out.shape[1:]))
    tail0 = ev(main) if timing is not None else None
    if xchg:
        main.wait_stream(side)
    if mark:
        mark("exchange_epilogue_tail")
    if timing is not None:
        tail1 = ev(main)
        timing.append(_LazyExchangeTiming(coll_events, tail0, tail1, bytes_reduced, n_groups))
    if rank != dst:
        return [None for _ in which]
    return result


class _LazyExchangeTiming(dict):
    """Exchange breakout of one sharded_measures call; the event arithmetic happens on first access (after a sync)."""

    def __init__(self, coll_events, tail0, tail1, bytes_reduced, n_groups):
        super().__init__(bytes_reduced=bytes_reduced, n_groups=n_groups)
        self._pending = (coll_events, tail0, tail1)

    def __missing__(self, key):
        coll_events, tail0, tail1 = self._pending
        tail1.synchronize()
        self["collective_ms"] = float(sum(a.elapsed_time(b) for a, b in coll_events))
        self["exposed_ms"] = float(tail0.elapsed_time(tail1))
        return self[key]


def total_observations_equal(local_n_obs, world):
    """n_observations of the job when every rank holds the same number of trials (no collective needed)."""
    return int(local_n_obs) * int(world)


# ---- a Connectivity whose trials live on several GPUs ------------------------------------------------------
def all_reduce_sum_(t, group=None):
    """In-place sum over ranks (gloo moves device tensors through the host)."""
    if _no_exchange(group):
        return t
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        host = t.cpu()
        dist.all_reduce(host, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, group=group)
    return t


def merge_disjoint(values, group=None):
    """Every rank filled a DISJOINT subset of the entries of a float64 array and left NaN elsewhere (channel pairs of
    the Granger prediction dealt out over the ranks): the union on every rank, NaN where nobody wrote.  One sum of the
    NaN-cleared values and one of the written-flags."""
    if _no_exchange(group):
        return values
    written = ~torch.isnan(values)
    vals = torch.where(written, values, torch.zeros_like(values))
    count = written.to(values.dtype)
    all_reduce_sum_(vals, group)
    all_reduce_sum_(count, group)
    return torch.where(count > 0, vals, torch.full_like(vals, float("nan")))


def deal(items, world_size, rank):
    """Round-robin share of a work list (pairs, (bin, group pair) tasks): cost per item is uniform, and a round-robin
    deal needs no knowledge of the list's length on the other ranks."""
    return items[rank::world_size]


def _connectivity_base():
    from .connectivity import Connectivity
    return Connectivity


class ShardedConnectivity(_connectivity_base()):
    """``Connectivity`` over trials that are spread across the GPUs of a node.

    Every rank (one process per GPU, ``torch.distributed`` initialised, backend ``nccl`` = RCCL) builds it from ITS
    trials -- ``ShardedConnectivity.from_multitaper(Multitaper(x[:, lo:hi]))`` with ``lo, hi = shard_bounds(R, world,
    rank)`` -- and calls the measures collectively; every rank gets the full result.

    * expectation measures (power ... pairwise_phase_consistency): local un-normalised records -> reduce-scatter
      over bins -> epilogue on the owned 1/N of the bins with the job-wide n_observations -> all-gather;
    * pairwise spectral Granger: records all-reduced once, the channel pairs dealt out over the ranks, each rank
      factorises its pairs, one merge of the disjoint outputs;
    * canonical coherence: records all-reduced once, the bins split over the ranks, all-gather of the owned bins;
    * the full Wilson factor / MVAR measures and global coherence: records all-reduced, then computed redundantly
      (one C x C problem per window: nothing to deal out below the window count).

    ``expectation_type`` must average over trials ("trials", "time_trials", "trials_tapers",
    "time_trials_tapers"): a kept trial axis would make the output itself sharded.
    """

    def __init__(self, *args, process_group=None, **kwargs):
        super().__init__(*args, **kwargs)
        if "trials" not in self.expectation_type.split("_"):
            raise ValueError("ShardedConnectivity shards the trials: expectation_type must average over them "
                             f"(got {self.expectation_type!r})")
        self._group = process_group
        self._world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self._rank = dist.get_rank(process_group) if self._world > 1 else 0
        self._sharded_cache = {}
        # trials of the whole job: ONE collective, at construction (which every rank takes part in anyway), so that no
        # later call -- whatever its expectation type, on whichever subset of the ranks -- needs another one
        self._n_trials_local = int(self._shape5[1])
        if self._n_trials_local < 1:
            raise ValueError("ShardedConnectivity needs at least one trial on every rank")
        self._n_trials_total = total_observations(self._n_trials_local, process_group)

    @classmethod
    def from_multitaper(cls, multitaper_instance, expectation_type="trials_tapers", blocks=None, dtype=None,
                        process_group=None):
        import numpy as np
        from . import options
        dtype = np.complex128 if dtype is None else dtype
        precision = options.engine_precision(dtype)
        if precision == "float32" and not np.iscomplexobj(multitaper_instance.time_series):
            from .connectivity import _PendingSpectra         # the transform runs at the first request (Connectivity.from_multitaper)
            multitaper_instance.check_device_path()
            first = _PendingSpectra(multitaper_instance, precision)
        else:
            first = multitaper_instance.device_spectra(precision=precision)
        obj = cls(first, expectation_type=expectation_type, time=multitaper_instance.time,
                  frequencies=multitaper_instance.frequencies, blocks=blocks, dtype=dtype, process_group=process_group)
        obj._multitaper = multitaper_instance
        return obj

    @property
    def n_observations(self):
        """Of the whole job (reference connectivity.py:594-610 over all trials)."""
        return self._n_observations_total(super().n_observations)

    def _n_observations_total(self, local_n_obs):
        """Every expectation type that averages over trials counts (trials) x (windows and / or tapers): the local count
        is the local trial count times a factor that is the same on every rank, whichever type the caller accumulated
        with (the measures use self.expectation_type, canonical / global coherence always trials x tapers)."""
        per_trial, rem = divmod(int(local_n_obs), self._n_trials_local)
        assert rem == 0, "n_observations of a trial-averaging expectation is a multiple of the trial count"
        return per_trial * self._n_trials_total

    def _reduce_over_ranks(self, accum):
        """Replicated sum of the records (Granger, canonical, MVAR, global coherence read every bin)."""
        return all_reduce_sum_(accum, self._group)

    def _measure(self, which):
        """Reduce-scatter over bins, epilogue on the owned bins, all-gather; the scattered shard is kept per plane
        set, so several measures share one exchange."""
        from . import _lib, engine
        planes = _lib.MEASURE_PLANES[which]
        key = None
        for have in self._sharded_cache:
            if isinstance(have, int) and have & planes == planes:
                key = have
                break
        if key is None:
            sp = self._device(planes_hint=planes)
            accum, n_obs = engine.accumulate(sp, self.expectation_type, planes, n_freq=self._n_freq)
            shard, lo, hi = reduce_scatter_bins(accum, self._group)
            key = planes
            self._sharded_cache[key] = (shard, accum.shape[0], self._n_observations_total(n_obs))
        shard, n_bins, n_total = self._sharded_cache[key]
        C = self._shape5[4]
        out = all_gather_bins(engine.measure(shard, C, key, n_total, which, wide=self._wide_output(which)), n_bins, self._group)
        tail = (C,) if which == _lib.M_POWER else (C, C)
        return engine.to_host(out).reshape(self._kept_shape() + (self._n_freq,) + tail)

    def _granger(self, pairs):
        """This rank's share of the pairs, then one merge of the disjoint outputs."""
        import numpy as np
        accum, _, n_freq = self._csm_records("granger")          # collective: every rank takes part, pairs or not
        mine = deal(np.asarray(pairs, dtype=np.int32).reshape(-1, 2), self._world, self._rank)
        N, C = self._shape5[3], self._shape5[4]
        if len(mine):
            out = self._granger_device(mine)
        else:
            out = torch.full((accum.shape[0] // n_freq, N // 2 + 1, C, C), float("nan"), dtype=torch.float64,
                             device=accum.device)
        merged = merge_disjoint(out, self._group)
        return engine_to_host(merged).reshape(self._kept_shape() + (N // 2 + 1, C, C))

    def _canonical_bins(self, n_bins):
        """The bins this rank evaluates: a contiguous 1/N, padded so that every rank holds the same count."""
        per = padded_bins(n_bins, self._world) // self._world
        lo = min(self._rank * per, n_bins)
        return lo, min(lo + per, n_bins), per

    def _canonical_gather(self, part, n_bins, per):
        if _no_exchange(self._group):
            return part
        if part.shape[0] < per:
            pad = torch.full((per - part.shape[0],) + tuple(part.shape[1:]), float("nan"), dtype=part.dtype,
                             device=part.device)
            part = torch.cat([part, pad], dim=0)
        if dist.get_backend(self._group) == "gloo" and part.is_cuda:
            return all_gather_bins(part.cpu(), n_bins, self._group).to(part.device)
        return all_gather_bins(part, n_bins, self._group)


def engine_to_host(t):
    from . import engine
    return engine.to_host(t)
