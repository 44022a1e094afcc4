"""Developer tool (GPU box): throughput of the standalone operators of the boundary (bh_radix_argsort, bh_prefix_sum,
bh_tile_sort_offsets) on random inputs — the reference's generic brush-sort / brush-prefix-sum calls, for callers that assemble the
forward themselves.  Prints ms per call (torch events over 20 calls after 3 warm-ups) and the bytes a pass structure moves.
    python scripts/operator_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import brush_amd as ba   # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    for rep in range(2):   # (the first round also pays the process's first-use costs: shown, not hidden)
        for n in (100_000, 1_000_000, 10_000_000, 30_000_000):
            keys = torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int32, device=dev, generator=g)
            vals = torch.arange(n, dtype=torch.int32, device=dev)
            for bits in (32, 13):
                k = keys if bits == 32 else (keys & ((1 << bits) - 1))
                ms = timed(lambda: ba.radix_argsort(k, vals, bits))
                print("round %d radix_argsort n %9d bits %2d: %.3f ms  (%.1f M keys/s)" % (rep, n, bits, ms, n / ms / 1e3), flush=True)
            ms = timed(lambda: ba.prefix_sum(keys & 0xFF))
            print("round %d prefix_sum    n %9d        : %.3f ms  (%.1f GB/s read + write)" % (rep, n, ms, 8 * n / ms / 1e6), flush=True)
            tiles = 8160
            tid = (keys & 0x7FFFFFFF) % tiles
            ms = timed(lambda: ba.tile_sort_offsets(tid, vals, tiles))
            print("round %d tile_sort_offsets n %9d tiles %d: %.3f ms  (%.1f M pairs/s)" % (rep, n, tiles, ms, n / ms / 1e3), flush=True)


if __name__ == "__main__":
    main()
