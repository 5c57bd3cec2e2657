"""Multi-GPU sharding of one frame: one process per GPU, no collective on the data path, ONE
exchange at the end onto rank 0 (RCCL over xGMI when the process group's backend is "nccl"):
sample ranges -> reduce(sum) of the per-pixel radiance sums, W * H * 24 bytes per rank; row bands ->
gather of every rank's own rows, packed, W * H * 24 / world bytes per rank (disjoint supports:
nothing to add, and the frame is the single-GPU frame bit for bit).

The per-pixel value of the reference is  sum_s final_s / spp  followed by one clamp
(src/tracing/cPathTracer.cpp:692,701-702 of the reference), i.e. linear in the samples before the
clamp.  So a frame can be cut either by sample ranges (every rank renders all pixels for its own
samples, images are SUMMED) or by interleaved row bands (disjoint supports, SUM is a union).  Random
numbers are keyed by the global (pixel, sample) index, so every cut renders the same samples."""
from __future__ import annotations


def shard_samples(spp_total, rank, world):
    """Contiguous sample range of `rank`: (offset, count); the ranges tile [0, spp_total)."""
    base, rem = divmod(int(spp_total), int(world))
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def shard_bands(rank, world, band=8):
    """Interleaved row bands: rank renders rows r with (r // band) % world == rank."""
    return dict(row_stride=int(world), row_phase=int(rank), band=int(band))


def band_rows(height, rank, world, band=8):
    """Rows of the frame that `rank` renders under shard_bands, ascending (a list of ints)."""
    return [r for r in range(int(height)) if (r // int(band)) % int(world) == int(rank)]


def gather_bands(out, rank, world, band=8):
    """The exchange of the bands scheme: every rank sends ITS rows of `out` ([H, W, 3] float64), packed
    into one message (padded to the largest share so that all messages have one size), rank 0 puts
    them in place.  W * H * 24 / world bytes per rank instead of the whole frame."""
    import torch
    import torch.distributed as dist
    H = out.shape[0]
    flat = out.view(H, -1)
    share = [band_rows(H, r, world, band) for r in range(world)]
    most = max(len(x) for x in share)
    packed = torch.zeros((most, flat.shape[1]), dtype=out.dtype, device=out.device)
    mine = torch.tensor(share[rank], dtype=torch.long, device=out.device)
    if len(share[rank]):
        packed[: len(share[rank])] = flat.index_select(0, mine)
    if rank == 0:
        parts = [torch.empty_like(packed) for _ in range(world)]
        dist.gather(packed, gather_list=parts, dst=0)
        for r in range(1, world):
            if len(share[r]):
                flat.index_copy_(0, torch.tensor(share[r], dtype=torch.long, device=out.device), parts[r][: len(share[r])])
    else:
        dist.gather(packed, dst=0)


def render_frame(render_fn, out, spp_total, rank=0, world=1, scheme="samples", band=8, reduce=True,
                 collective=None, reduce_ms=None, **kw):
    """Renders this rank's share into `out` (a torch tensor [H,W,3] float64 on the rank's device)
    and reduces onto rank 0.  `render_fn(out, **options)` fills `out` with pre-clamp sums; for the
    GPU core it is `lambda out, **o: ctx.render_device(out.data_ptr(), **o)`.  `collective`: run
    the reduce even in a world of one (default: only when world > 1).  reduce_ms: a list that
    receives the duration of the exchange in milliseconds (events on the tensor's stream; host clock
    for CPU tensors): the one step of a frame that is not rendering."""
    if scheme == "samples":
        offset, count = shard_samples(spp_total, rank, world)
        if count > 0:
            stats = render_fn(out, spp=count, sample_offset=offset, spp_total=spp_total, **kw)
        else:
            out.zero_()
            stats = None
    elif scheme == "bands":
        stats = render_fn(out, spp=spp_total, sample_offset=0, spp_total=spp_total, **shard_bands(rank, world, band), **kw)
    else:
        raise ValueError("scheme must be 'samples' or 'bands'")
    if reduce and (world > 1 if collective is None else collective):
        import torch.distributed as dist

        def exchange():
            if scheme == "bands":
                gather_bands(out, rank, world, band)
            else:
                dist.reduce(out, dst=0, op=dist.ReduceOp.SUM)

        if reduce_ms is None:
            exchange()
        elif out.is_cuda:
            import torch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            exchange()
            e1.record()
            e1.synchronize()
            reduce_ms.append(e0.elapsed_time(e1))
        else:
            import time
            t0 = time.perf_counter()
            exchange()
            reduce_ms.append((time.perf_counter() - t0) * 1e3)
    return stats
