"""N > 1 path on CPU: world_size-2 gloo process group (no GPU needed).
Checks the collective semantics the data-parallel train step relies on (SUM of the leading
part of the exchange buffer, the rest untouched; MAX of the RefineRecord maxima before refine;
identical result on every rank) and the view sharding used by bench.py / SplatTrainer."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from brush_amd.parallel import allreduce_exchange, allreduce_refine_maxima, view_for_rank
    n, C = 1000, 4
    g = torch.Generator().manual_seed(100 + rank)
    exch = torch.randn(n * (10 + 3 * C + 3), generator=g)    # visible | grads | refine
    sum_count = n * (10 + 3 * C + 2)                          # cameras mode: refine stays local
    norm, screen = torch.rand(n, generator=g), torch.rand(n, generator=g)
    mine_e, mine_n, mine_s = exch.clone(), norm.clone(), screen.clone()
    w = allreduce_exchange(exch, sum_count)
    allreduce_refine_maxima(norm, screen)
    views = [view_for_rank(s, rank, world, 8) for s in range(4)]
    q.put((rank, w, sum_count, mine_e.numpy(), mine_n.numpy(), mine_s.numpy(), exch.numpy(), norm.numpy(), screen.numpy(), views))
    dist.destroy_process_group()


def test_allreduce_semantics_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w0, sc, e0, n0, s0, re0, rn0, rs0, v0), (_, w1, _, e1, n1, s1, re1, rn1, rs1, v1) = res
    assert w0 == w1 == 2
    assert np.array_equal(re0[:sc], re1[:sc]), "every rank holds the same reduced gradients"
    assert np.allclose(re0[:sc], e0[:sc] + e1[:sc], rtol=0, atol=1e-6)
    assert np.array_equal(re0[sc:], e0[sc:]) and np.array_equal(re1[sc:], e1[sc:]), "the tail (refine weight) stays rank-local"
    assert np.array_equal(rn0, np.maximum(n0, n1)) and np.array_equal(rn0, rn1)
    assert np.array_equal(rs0, np.maximum(s0, s1)) and np.array_equal(rs0, rs1)
    # one pass over 8 views with 2 ranks x 4 steps visits every view exactly once
    assert sorted(v0 + v1) == list(range(8))


def test_view_sharding_properties():
    from brush_amd.parallel import view_for_rank
    for world in (1, 2, 4, 8):
        seen = [view_for_rank(s, r, world, 16) for s in range(16 // world) for r in range(world)]
        assert sorted(seen) == list(range(16))
    assert view_for_rank(5, 0, 1, 3) == 2


# ---- one frame partitioned by strips of tile rows (BASELINE.json configs[4]) ---------------
def _strip_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from brush_amd.parallel import allgather_strips, strip_spans_px
    h, w = 150, 37   # 10 tile rows -> strips of 5 rows; the last strip is cut by the image edge
    full = torch.arange(h * w * 4, dtype=torch.float32).reshape(h, w, 4)
    spans = strip_spans_px(h, world)
    b, e = spans[rank]
    img = torch.full((h, w, 4), -1.0)
    img[b:e] = full[b:e]
    a = img.clone()
    allgather_strips(a, b, e, spans=spans)
    c = img.clone()
    allgather_strips(c, b, e)   # spans exchanged by a collective
    q.put((rank, bool(torch.equal(a, full)), bool(torch.equal(c, full)), spans))
    dist.destroy_process_group()


def test_allgather_strips_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_strip_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, ok_a, ok_c, spans in res:
        assert ok_a and ok_c
        assert spans == [(0, 80), (80, 150)]


def _halo_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from brush_amd.parallel import STRIP_HALO_PX, exchange_strip_halos, strip_spans_px, strips_allow_halo_loss
    h, w = 150, 23   # 10 tile rows over 3 ranks: strips of 4 / 3 / 3 tile rows, the last cut to 22 px by the image edge
    full = torch.arange(h * w * 4, dtype=torch.float32).reshape(h, w, 4)
    spans = strip_spans_px(h, world)
    b, e = spans[rank]
    img = torch.full((h, w, 4), -1.0)
    img[b:e] = full[b:e]
    exchange_strip_halos(img, spans, rank)
    lo, hi = max(b - STRIP_HALO_PX, 0), min(e + STRIP_HALO_PX, h)
    ok = bool(torch.equal(img[lo:hi], full[lo:hi]))                       # strip + both halos hold the frame's rows
    untouched = bool((img[:lo] == -1).all() and (img[hi:] == -1).all())   # nothing else was written
    q.put((rank, ok, untouched, spans, strips_allow_halo_loss(spans)))
    dist.destroy_process_group()


def test_strip_halo_exchange_world3():
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_halo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, ok, untouched, spans, allowed in res:
        assert ok and untouched and allowed
        assert spans == [(0, 64), (64, 112), (112, 150)]
    from brush_amd.parallel import strips_allow_halo_loss
    assert not strips_allow_halo_loss([(0, 16), (16, 112)])    # a 1-tile-row strip cannot serve a 21-px halo


def test_tile_row_partition_properties():
    from brush_amd.parallel import tile_rows_for_rank
    for tile_bh in (1, 7, 68, 135):
        for world in (1, 2, 3, 8):
            if world > tile_bh:
                with pytest.raises(ValueError):
                    tile_rows_for_rank(tile_bh, 0, world)
                continue
            rows = [tile_rows_for_rank(tile_bh, r, world) for r in range(world)]
            assert rows[0][0] == 0 and rows[-1][1] == tile_bh
            assert all(a[1] == b[0] for a, b in zip(rows, rows[1:]))       # contiguous, disjoint
            sizes = [e - b for b, e in rows]
            assert min(sizes) >= 1 and max(sizes) - min(sizes) <= 1
    # weighted cuts follow the work, keep every rank non-empty and stay a partition
    w = [1.0] * 4 + [10.0] * 2 + [1.0] * 4
    rows = [tile_rows_for_rank(10, r, 3, w) for r in range(3)]
    assert rows[0][0] == 0 and rows[-1][1] == 10 and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
    assert all(e > b for b, e in rows)
    loads = [sum(w[b:e]) for b, e in rows]
    assert max(loads) <= 0.6 * sum(w)


def _direct_worker(rank, world, port, q, counts):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from brush_amd.parallel import allreduce_direct, direct_chunk
    out = []
    for count in counts:
        g = torch.Generator().manual_seed(7 * count + rank)
        buf = torch.randn(count + 13, generator=g)
        mine = buf.clone()
        allreduce_direct(buf, count)
        out.append((mine.numpy(), buf.numpy()))
        chunks = [direct_chunk(count, world, c) for c in range(world)]
        assert chunks[0][0] == 0 and all(chunks[c][0] + chunks[c][1] == chunks[c + 1][0] or chunks[c + 1][1] == 0 for c in range(world - 1))
        assert sum(n for _, n in chunks) == count and all(b % 4 == 0 or n == 0 for b, n in chunks)
    q.put((rank, out))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 5])
def test_direct_allreduce_reduce_scatter_allgather_over_p2p(world):
    """parallel.allreduce_direct restates the library's direct all-reduce (comm.hip: every rank sends chunk p to rank p, the owner
    adds the versions up in rank order and sends the result to everybody): equal on every rank BIT FOR BIT, equal to the sum in rank
    order, nothing behind `count` touched; ragged and empty tail chunks (count = 5 over 5 ranks: chunks of 4, 1, 0, 0, 0)."""
    counts = [70003, 65536, 5, 1, 4 * world, 4 * world + 1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_direct_worker, args=(r, world, port, q, counts)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for k, count in enumerate(counts):
        want = np.zeros(count, np.float32)
        for r in range(world):
            want = want + res[r][1][k][0][:count]   # float32, rank order
        for r in range(world):
            mine, got = res[r][1][k]
            assert np.array_equal(got[:count], want), (world, count, r)
            assert np.array_equal(got[count:], mine[count:])
