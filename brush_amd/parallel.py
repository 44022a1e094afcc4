"""Multi-GPU helpers (one process per GPU, torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" in the CPU tests).

The reference is single-GPU, one view per step (brush-train/src/train.rs:176-186): data
parallelism over cameras is a capability this build adds.  Semantics:
every rank holds a full replica of the splats + Adam state and renders its own view; between
backward and Adam the step's ONE exchange buffer (visible flags | fused gradients) is
SUM-all-reduced — a single large message per step, the right shape for xGMI's per-link bound —
and the gradients are scaled by 1/world inside the update (mean gradient over the K views), so
every rank applies the identical update.  vis_weight counts the views that saw a splat.  The two
running maxima of the RefineRecord (refine_weight_norm, max_screen_size) stay rank-local between
refines and are MAX-reduced once, right before refine (`allreduce_refine_maxima`) — max is
associative, so this equals reducing every step at 1/refine_every of the traffic.
K = 1 is bit-identical to the single-GPU step.
"""
import torch


def view_for_rank(step, rank, world, num_views):
    """Round-robin sharding of a view list over ranks: at `step` rank r takes view
    (step * world + r) mod num_views, so one pass over the dataset visits every view once
    when num_views is a multiple of world."""
    return (step * world + rank) % num_views


def allreduce_exchange(exchange: torch.Tensor, sum_count: int, group=None):
    """In place: the first `sum_count` floats of the step's exchange buffer <- sum over ranks
    (bh_grad_hook contract, include/brush_hip.h).  At 1 M splats / SH0 that is 60 MB, 240 MB at SH3."""
    import torch.distributed as dist
    dist.all_reduce(exchange[:sum_count], op=dist.ReduceOp.SUM, group=group)
    return dist.get_world_size(group)


DIRECT_ALLREDUCE_MIN_FLOATS = 1 << 16   # as brush_amd/csrc/context.h: shorter messages stay with all_reduce


def direct_chunk(count, world, c):
    """[begin, begin + len) of chunk c when `count` floats are cut for `world` ranks — brush_amd/csrc/comm.hip direct_chunk:
    ceil(count / world) rounded up to 4 floats per chunk, the tail ragged (or empty)."""
    per = ((count + world - 1) // world + 3) & ~3
    b = min(c * per, count)
    return b, (min(per, count - b) if b < count else 0)


def allreduce_direct(exchange: torch.Tensor, sum_count: int, group=None):
    """The library's direct all-reduce (comm.hip comm_allreduce_direct: reduce-scatter + all-gather over point-to-point messages,
    for the fully connected xGMI node — every rank sends chunk p to rank p, the owner of a chunk adds the `world` versions up IN
    RANK ORDER, then sends the finished chunk to everybody) restated over torch.distributed isend / irecv: what the exchange hook
    runs with SplatTrainer(allreduce="direct"), and what rehearses the algorithm where RCCL cannot run (gloo, ranks sharing a GPU).
    In place on exchange[:sum_count]; the result is the same on every rank bit for bit."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world == 1 or sum_count == 0:
        return world
    flat = exchange[:sum_count]
    staged = flat.is_cuda and dist.get_backend(group) == "gloo"   # (gloo moves host memory)
    t = flat.cpu() if staged else flat
    my_b, my_n = direct_chunk(sum_count, world, rank)
    per = direct_chunk(sum_count, world, 0)[1]
    scratch = torch.empty((world - 1, max(per, 1)), dtype=t.dtype, device=t.device)
    peer = lambda p: p if group is None else dist.get_global_rank(group, p)   # noqa: E731
    ops = []
    for p in range(world):
        if p == rank:
            continue
        pb, pn = direct_chunk(sum_count, world, p)
        if pn:
            ops.append(dist.P2POp(dist.isend, t[pb:pb + pn], peer(p), group))
        if my_n:
            ops.append(dist.P2POp(dist.irecv, scratch[p - (1 if p > rank else 0), :my_n], peer(p), group))
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    if my_n:
        acc = torch.zeros(my_n, dtype=t.dtype, device=t.device)
        for p in range(world):   # rank order, like reduce_versions_kernel
            acc += t[my_b:my_b + my_n] if p == rank else scratch[p - (1 if p > rank else 0), :my_n]
        t[my_b:my_b + my_n] = acc
    ops = []
    for p in range(world):
        if p == rank:
            continue
        pb, pn = direct_chunk(sum_count, world, p)
        if my_n:
            ops.append(dist.P2POp(dist.isend, t[my_b:my_b + my_n], peer(p), group))
        if pn:
            ops.append(dist.P2POp(dist.irecv, t[pb:pb + pn], peer(p), group))
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    if staged:
        flat.copy_(t)
    return world


def allreduce_refine_maxima(refine_weight_norm: torch.Tensor, max_screen_size: torch.Tensor, group=None):
    """In place: both running maxima <- max over ranks (one collective on a fused staging tensor).
    Called before refine, so every rank takes the identical prune / split decisions."""
    import torch.distributed as dist
    n = refine_weight_norm.numel()
    both = torch.cat([refine_weight_norm.reshape(-1), max_screen_size.reshape(-1)])
    dist.all_reduce(both, op=dist.ReduceOp.MAX, group=group)
    refine_weight_norm.copy_(both[:n].view_as(refine_weight_norm))
    max_screen_size.copy_(both[n:].view_as(max_screen_size))


# ---------------------------------------------------------------------------
# One frame partitioned over GPUs by strips of tile rows (BASELINE.json configs[4])
# ---------------------------------------------------------------------------
def tile_rows_for_rank(tile_bh, rank, world, weights=None):
    """[begin, end) tile rows of `rank`: contiguous strips, sizes differing by at most one row.
    With `weights` (per-row work, e.g. last frame's intersections per tile row) the cuts balance
    the prefix sum of the weights instead of the row count."""
    if world > tile_bh:
        raise ValueError("more ranks (%d) than tile rows (%d)" % (world, tile_bh))
    if weights is None:
        base, extra = divmod(tile_bh, world)
        begin = rank * base + min(rank, extra)
        return begin, begin + base + (1 if rank < extra else 0)
    w = [float(x) for x in weights]
    if len(w) != tile_bh:
        raise ValueError("weights must have one entry per tile row")
    total = sum(w) or 1.0
    cuts, acc, r = [0], 0.0, 1
    for row, x in enumerate(w):
        acc += x
        # keep at least one row for every remaining rank
        while r < world and acc >= total * r / world and row + 1 <= tile_bh - (world - r) and row + 1 > cuts[-1]:
            cuts.append(row + 1)
            r += 1
    while len(cuts) < world:
        cuts.append(max(cuts[-1] + 1, tile_bh - (world - len(cuts))))
    cuts.append(tile_bh)
    return cuts[rank], cuts[rank + 1]


def strip_spans_px(img_h, world, weights=None):
    """Pixel-row span [begin, end) of every rank's strip (host arithmetic, identical on all ranks)."""
    tile_bh = (img_h + 15) // 16
    out = []
    for r in range(world):
        b, e = tile_rows_for_rank(tile_bh, r, world, weights)
        out.append((b * 16, min(e * 16, img_h)))
    return out


def allgather_strips(img: torch.Tensor, row_begin_px: int, row_end_px: int, group=None, spans=None):
    """In place on img [H,W,C]: this rank owns pixel rows [row_begin_px, row_end_px); afterwards
    every rank holds the whole image.  Strips are contiguous in the HWC layout, so each is one
    message; heights may differ by a tile row, so strips are padded to the tallest one and moved
    with a single all_gather_into_tensor."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return
    h = img.shape[0]
    row_elems = img[0].numel()
    if spans is None:  # exchange the spans (one small collective + a host read); callers that
        # partition with tile_rows_for_rank pass strip_spans_px(...) and skip this
        mine = torch.tensor([row_begin_px, row_end_px], dtype=torch.int64, device=img.device)
        got = torch.empty(world * 2, dtype=torch.int64, device=img.device)
        dist.all_gather_into_tensor(got, mine, group=group)
        spans = [tuple(x) for x in got.view(world, 2).tolist()]
    tallest = max(e - b for b, e in spans)
    send = torch.zeros(tallest * row_elems, dtype=img.dtype, device=img.device)
    send[: (row_end_px - row_begin_px) * row_elems] = img[row_begin_px:row_end_px].reshape(-1)
    recv = torch.empty(world * tallest * row_elems, dtype=img.dtype, device=img.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, tallest * row_elems)
    flat = img.view(h, row_elems)
    for r, (b, e) in enumerate(spans):
        if r != dist.get_rank(group):
            flat[b:e] = recv[r, : (e - b) * row_elems].view(e - b, row_elems)


STRIP_HALO_PX = 21  # one 16-px tile row + the 5-px reach of the 11-tap SSIM window (brush_amd/csrc/loss_fused.hip)


def exchange_strip_halos(img: torch.Tensor, spans, rank: int, group=None, halo: int = STRIP_HALO_PX):
    """In place on img [H,W,C]: fetch the `halo` pixel rows just above and just below this rank's strip from the
    neighbouring ranks (strip-wise loss, SURVEY.md §8e: "exchange 5 rows with neighbours" — 21 here because the loss
    kernels work in whole tile rows).  One all_gather_into_tensor of [2, halo, W, C] per rank instead of the whole
    image.  Every strip must be at least `halo` rows tall (callers fall back to allgather_strips otherwise)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return
    h = img.shape[0]
    b, e = spans[rank]
    send = torch.zeros((2, halo) + tuple(img.shape[1:]), dtype=img.dtype, device=img.device)
    send[0, : min(halo, e - b)] = img[b: min(b + halo, e)]
    send[1, halo - min(halo, e - b):] = img[max(e - halo, b): e]
    recv = torch.empty(world * send.numel(), dtype=img.dtype, device=img.device)
    dist.all_gather_into_tensor(recv, send.view(-1), group=group)
    recv = recv.view((world,) + tuple(send.shape))
    if rank > 0:                       # rows [b - halo, b) = the last rows of the strip above
        k = min(halo, b)
        img[b - k: b] = recv[rank - 1, 1, halo - k:]
    if rank < world - 1 and e < h:     # rows [e, e + halo) = the first rows of the strip below
        k = min(halo, h - e, spans[rank + 1][1] - spans[rank + 1][0])
        img[e: e + k] = recv[rank + 1, 0, :k]


def strips_allow_halo_loss(spans, halo: int = STRIP_HALO_PX):
    """The strip-wise loss needs every neighbour to own at least `halo` rows (else a halo would span two ranks)."""
    return all(e - b >= halo for b, e in spans)
