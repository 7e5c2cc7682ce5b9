"""Position-range sharding of the scan over the GPUs of one node.

The reference partitions positions [0,N) over OpenMP threads (src/ClusterLCP.cpp:150-161): a
thread skips to the first cluster head in its chunk (:196-202) and reads past its end until
the open run closes (:246-264).  Here a rank owns a tile-aligned range [lo,hi) and holds a
read-ahead halo [hi,hi_halo); a cluster belongs to the rank that owns its first position.
Per-rank outputs: a private uint8 table (combined by ONE all-reduce, sum modulo 256), the
cluster count (sum) and the maximum length (max).  No other data-path collective.
"""
from __future__ import annotations

from ._lib import MAX_CLUSTER, TILE

DEFAULT_HALO = MAX_CLUSTER + TILE      # any run the scorer accepts closes inside it


def shard_ranges(n, world, halo=DEFAULT_HALO):
    """-> list of (lo, hi, hi_halo) per rank; cuts are multiples of the 4096-position tile."""
    tiles = (n + TILE - 1) // TILE
    out = []
    for r in range(world):
        t0 = tiles * r // world
        t1 = tiles * (r + 1) // world
        lo, hi = min(t0 * TILE, n), min(t1 * TILE, n)
        if r == world - 1:
            hi = n
        out.append((lo, hi, min(hi + halo, n) if hi < n else n))
    return out


def allreduce_tables(sim_t, group=None):
    """Sum the per-rank uint8 tables modulo 256 in place (one collective: RCCL on GPUs)."""
    import torch.distributed as dist
    dist.all_reduce(sim_t, op=dist.ReduceOp.SUM, group=group)
    return sim_t


def combine_counters(n_clusters, max_len, device, group=None):
    """(sum of cluster counts, max of maximum lengths) over ranks."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([n_clusters], dtype=torch.int64, device=device)
    m = torch.tensor([max_len], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
    return int(t.item()), int(m.item())


def check_uint8_sum_wraps(device, group=None):
    """Self-check that the backend's uint8 SUM wraps modulo 256 (200 + 100 -> 44 for 2 ranks)."""
    import torch
    import torch.distributed as dist
    w = dist.get_world_size(group)
    t = torch.full((64,), 200, dtype=torch.uint8, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    want = (200 * w) % 256
    if not bool((t == want).all()):
        raise RuntimeError(f"uint8 all-reduce does not wrap modulo 256: got {int(t[0])}, want {want}")
