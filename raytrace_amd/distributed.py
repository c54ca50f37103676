"""Multi-GPU partitioning of a frame: image tiles x sample ranges, one process per GPU.

Every path is independent, so a frame splits two ways: by pixel rows (disjoint pixels, nothing to combine but a
gather) and by sample range (same pixels; the per-GPU FP32 accumulators must be summed).  With the counter RNG a
sample's random stream depends on (seed, pixel, sample) only, so the image does not depend on the split.  The sum
is one collective on the accumulator tensor: `torch.distributed.reduce` to the root, which on ROCm's "nccl"
backend is RCCL over xGMI (the root receives world-1 partial images over its point-to-point links; nobody else
needs the result, so no all-reduce).

State of the evidence: on hardware the device-ordered exchange (events between the library's stream and the
collective's, the cached send / receive buffers of gather_rows, the in-place reduce) has run with ONE rank only; with two
and three ranks it has run through gloo on the CPU and on ranks sharing one GPU (tests/test_distributed.py).
"""
from dataclasses import dataclass


@dataclass(frozen=True)
class Share:
    """What one rank renders: rows [y0,y1) of the image — all of them, or with strips > 0 the strips of eight rows
    whose number is `strip` modulo `strips` — and samples [s0,s1) of each of their pixels."""
    y0: int
    y1: int
    s0: int
    s1: int
    strips: int = 0
    strip: int = 0


def split_frame(height: int, samples: int, world: int, row_groups: int = 1):
    """Shares of `world` ranks: `row_groups` bands of rows, each band's samples split over world/row_groups ranks."""
    if world < 1 or row_groups < 1 or world % row_groups:
        raise ValueError("world must be a positive multiple of row_groups")
    per_band = world // row_groups
    if samples < per_band:
        raise ValueError("fewer samples than ranks per row band")
    shares = []
    for rank in range(world):
        band, part = divmod(rank, per_band)
        y0 = height * band // row_groups
        y1 = height * (band + 1) // row_groups
        s0 = samples * part // per_band
        s1 = samples * (part + 1) // per_band
        shares.append(Share(y0, y1, s0, s1))
    return shares


def split_interleaved(height: int, samples: int, world: int, sample_groups: int = 1):
    """Shares of `world` ranks for a frame whose rows differ in cost (sky above, spheres below): every rank takes
    every (world / sample_groups)-th strip of eight rows over the whole height, and one of `sample_groups` sample
    ranges.  Disjoint pixels per sample range, about equal cost per rank."""
    if world < 1 or sample_groups < 1 or world % sample_groups:
        raise ValueError("world must be a positive multiple of sample_groups")
    strips = world // sample_groups
    if samples < sample_groups:
        raise ValueError("fewer samples than sample groups")
    shares = []
    for rank in range(world):
        part, strip = divmod(rank, strips)
        s0 = samples * part // sample_groups
        s1 = samples * (part + 1) // sample_groups
        shares.append(Share(0, height, s0, s1, strips if strips > 1 else 0, strip if strips > 1 else 0))
    return shares


def apply_share(params, share: Share):
    """Copy of an rt_params restricted to a share (tile rows, strips and sample range)."""
    p = params.copy()
    p.tile_y0, p.tile_y1 = share.y0, share.y1
    p.sample_begin, p.sample_end = share.s0, share.s1
    p.strip_count, p.strip_index = share.strips, share.strip
    return p


def combine(accum, dst: int = 0, group=None, after=None, async_op=False):
    """Sum the ranks' accumulators (a torch tensor, device or CPU) onto rank `dst`.

    after: a Renderer whose queued work writes `accum`.  The sum is ordered behind it on the device — an event on the
    renderer's stream that torch's current stream waits for — with no host wait in between."""
    import torch.distributed as dist
    device_ordered = accum.is_cuda and dist.get_backend(group) == "nccl"
    if after is not None and accum.is_cuda:
        if device_ordered:
            import torch
            ev = torch.cuda.Event()
            ev.record(torch.cuda.ExternalStream(after.stream()))
            torch.cuda.current_stream().wait_event(ev)
        else:
            after.synchronize()  # gloo stages device tensors through the host on streams of its own: wait here
    work = dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group, async_op=async_op and device_ordered)
    return work if device_ordered else None


def band_rows(height: int, bands: int, strips: int):
    """Cut [0, height) into at most `bands` runs of rows whose starts are multiples of 8 * max(strips, 1), so that a
    share's strips of eight keep their numbers inside every band."""
    unit = 8 * max(strips, 1)
    units = (height + unit - 1) // unit
    bands = max(1, min(bands, units))
    cuts = [min(height, unit * (units * b // bands)) for b in range(bands + 1)]
    return [(cuts[b], cuts[b + 1]) for b in range(bands) if cuts[b + 1] > cuts[b]]


def render_and_combine(renderer, params, accum, dst: int = 0, bands: int = 4, group=None):
    """One frame share, rendered band by band into `accum` (float32 [H, W, 3] on the renderer's device, already set as
    its accumulator) and summed onto rank `dst` band by band: the collective of band k is queued behind the render
    of band k on the device and runs beside the render of band k + 1.  Returns when everything is queued; the
    caller's stream (torch's current one) is ordered behind the last sum, and the renderer's stream behind it too,
    so that the next frame's clear cannot overtake the sums."""
    import torch
    works = []
    for y0, y1 in band_rows(params.tile_y1 - params.tile_y0, bands, params.strip_count):
        p = params.copy()
        p.tile_y0, p.tile_y1 = params.tile_y0 + y0, params.tile_y0 + y1
        renderer.render(p)
        works.append(combine(accum[p.tile_y0:p.tile_y1], dst=dst, group=group, after=renderer, async_op=True))
    for w in works:
        if w is not None:
            w.wait()  # orders torch's current stream behind the collective (no host wait for device tensors)
    if accum.is_cuda:
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream())
        torch.cuda.ExternalStream(renderer.stream()).wait_event(done)


def share_rows(share: Share):
    """The image rows a share covers, in increasing order."""
    rows = range(share.y0, share.y1)
    if share.strips:
        rows = [y for y in rows if ((y - share.y0) // 8) % share.strips == share.strip]
    return list(rows)


_gather_cache = {}


def gather_rows(accum, shares, rank: int, dst: int = 0, group=None, after=None):
    """Shares with disjoint rows and the whole sample range each (strips, row bands): nothing has to be summed, so
    every rank sends the root only the rows it rendered — 1/world of the frame instead of all of it — and the root
    puts them in place.  `accum`: float32 [H, W, 3], device or CPU; `shares`: the shares of all ranks of the group,
    the same list on every rank.  Ordered on the device like combine(): behind `after`'s queued work, no host wait
    with the nccl backend; `after`'s stream is ordered behind the exchange, so its next clear cannot overtake it."""
    import torch
    import torch.distributed as dist
    world = len(shares)
    # a list that does not match the group would leave some rank waiting in the collective for ever: fail here instead
    if world != dist.get_world_size(group):
        raise ValueError(f"gather_rows: {world} shares for a group of {dist.get_world_size(group)} ranks")
    if rank != dist.get_rank(group):
        raise ValueError(f"gather_rows: called as rank {rank} by rank {dist.get_rank(group)} of the group")
    if any((s.s0, s.s1) != (shares[0].s0, shares[0].s1) for s in shares):
        raise ValueError("gather_rows needs shares that differ in rows only")
    device_ordered = accum.is_cuda and dist.get_backend(group) == "nccl"
    key = (tuple(shares), tuple(accum.shape), str(accum.device), rank)
    if key not in _gather_cache:
        rows = [torch.tensor(share_rows(s), dtype=torch.long, device=accum.device) for s in shares]
        seen = torch.cat(rows)
        if seen.numel() != torch.unique(seen).numel():
            raise ValueError("gather_rows needs disjoint shares")
        most = max(int(r.numel()) for r in rows)
        on_host = not device_ordered  # gloo: the exchange goes through host memory
        send = torch.zeros((most,) + tuple(accum.shape[1:]), dtype=accum.dtype, device="cpu" if on_host else accum.device)
        recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
        _gather_cache[key] = (rows, send, recv)
    rows, send, recv = _gather_cache[key]
    if after is not None and accum.is_cuda:
        if device_ordered:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.ExternalStream(after.stream()))
            torch.cuda.current_stream().wait_event(ev)
        else:
            after.synchronize()
    mine = rows[rank]
    send[:mine.numel()].copy_(accum.index_select(0, mine))
    dist.gather(send, gather_list=recv, dst=dst, group=group)
    if rank == dst:
        for q in range(world):
            if q != dst and rows[q].numel():
                accum.index_copy_(0, rows[q], recv[q][:rows[q].numel()].to(accum.device))
    if after is not None and accum.is_cuda and device_ordered:
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream())
        torch.cuda.ExternalStream(after.stream()).wait_event(done)
    elif accum.is_cuda:
        torch.cuda.current_stream().synchronize()
