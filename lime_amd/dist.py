"""Position-range sharding of the scan over the GPUs of one node.

The reference partitions positions [0,N) over OpenMP threads (src/ClusterLCP.cpp:150-161): a
thread skips to the first cluster head in its chunk (:196-202) and reads past its end until
the open run closes (:246-264).  Here a rank owns a tile-aligned range [lo,hi) and holds a
read-ahead halo [hi,hi_halo); a cluster belongs to the rank that owns its first position.
Per-rank outputs: a private uint8 table (combined by ONE all-reduce or reduce-scatter, sum modulo 256), the
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


def choose_exchange(max_updates_per_rank, sim_bytes):
    """dense (reduce-scatter of the ranks' byte tables: sim_bytes (G-1)/G over the links per rank, and every rank writes a
    whole table) or sparse (owner-partitioned exchange of 4-byte update records, lime_comm_exchange_records: 4 x updates
    (G-1)/G per rank, a rank writes its block only): whichever moves fewer bytes.  All ranks must call it with the same
    numbers (the largest update count of any rank)."""
    return "sparse" if 4 * int(max_updates_per_rank) < int(sim_bytes) else "dense"


def table_block_bytes(sim_bytes, world):
    """bytes of one rank's block when the table is cut into `world` equal, 16-byte aligned blocks
    (the buffer handed to reduce_scatter_tables must hold world * this many bytes, zero padded)"""
    return (sim_bytes + 16 * world - 1) // (16 * world) * 16


def reduce_scatter_tables(sim_t, out_t, group=None, async_op=False):
    """Sum the per-rank uint8 tables modulo 256 and leave rank r with block r of the result (the
    form SURVEY 8e prefers: clusterChoose is row-independent, so each rank goes on with its block and
    only half the bytes of an all-reduce cross xGMI).  len(sim_t) == world * len(out_t).
    Returns the async work handle (RCCL) or None."""
    import torch.distributed as dist
    if dist.get_backend(group) == "gloo":                 # CPU tests: gloo has no reduce-scatter
        dist.all_reduce(sim_t, op=dist.ReduceOp.SUM, group=group)
        out_t.copy_(sim_t.view(dist.get_world_size(group), -1)[dist.get_rank(group)])
        return None
    return dist.reduce_scatter_tensor(out_t, sim_t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def combine_edges(edges):
    """edges[k] = lime_stats_t.edge of shard k (position order): raises LimeError(LIME_ERR_MAXLEN) if a run that
    crosses shard borders is a read+genome cluster (the reference refuses it too), else returns None"""
    import numpy as np
    from . import _lib
    e = np.ascontiguousarray(edges, dtype=np.uint32)
    _lib.check(_lib.load().lime_combine_edges(e.ctypes.data, len(e)))


EDGE_OPEN = 8


class Comm:
    """The exchange step through the C ABI (include/lime_hip.h, lime_comm_*): RCCL ncclReduceScatter / ncclAllReduce
    on ncclUint8 with ncclSum, one rank per GPU.  torch.distributed is the control plane only: it carries the
    128-byte ncclUniqueId from rank 0 to the others, the timing maximum and the barriers."""

    def __init__(self, rank, world, device, group=None):
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import _lib
        self.lib, self.rank, self.world, self.device, self.group = _lib.load(), rank, world, device, group
        idbuf = (C.c_uint8 * 128)()
        if rank == 0:
            self._check(self.lib.lime_comm_unique_id(idbuf))
        t = torch.tensor(list(bytes(idbuf)), dtype=torch.uint8, device=device if dist.get_backend(group) == "nccl" else "cpu")
        dist.broadcast(t, src=0, group=group)
        idb = (C.c_uint8 * 128)(*t.cpu().tolist())
        h = C.c_void_p()
        self._check(self.lib.lime_comm_init(idb, rank, world, C.byref(h)))
        self.h = h

    def _check(self, rc):
        if rc:
            from ._lib import LimeError
            raise LimeError(rc, self.lib.lime_comm_error().decode(errors="replace"))

    def reduce_scatter_tables(self, sim_t, blk_t, blk_bytes, stream=None):
        """rank r receives block r (blk_bytes bytes) of the sum modulo 256 of everybody's sim_t (world * blk_bytes bytes)"""
        self._check(self.lib.lime_comm_reduce_scatter_tables(self.h, sim_t.data_ptr(), blk_t.data_ptr(), blk_bytes, stream))

    def allreduce_tables(self, sim_t, stream=None):
        self._check(self.lib.lime_comm_allreduce_tables(self.h, sim_t.data_ptr(), sim_t.numel(), stream))

    def exchange_records(self, ctx, n_reads, n_refs, block_t, stream=None):
        """owner-partitioned exchange of the update records ctx.fused_records_dev left (after ctx.stats()): this rank ends
        with bytes [cell_lo, cell_lo + block_bytes) of the finished table in block_t; returns (cell_lo, block_bytes)"""
        import ctypes as C
        lo, nb = C.c_uint64(0), C.c_uint64(0)
        self._check(self.lib.lime_comm_exchange_records(self.h, ctx.h, n_reads, n_refs, block_t.data_ptr(), block_t.numel(),
                                                        C.byref(lo), C.byref(nb), stream))
        return int(lo.value), int(nb.value)

    def combine_counters(self, n_clusters, max_len):
        import torch
        t = torch.tensor([n_clusters, max_len], dtype=torch.int64, device=self.device)
        self._check(self.lib.lime_comm_combine_counters(self.h, t.data_ptr(), torch.cuda.current_stream().cuda_stream))
        c, m = t.cpu().tolist()
        return int(c), int(m)

    def check_uint8_sum_wraps(self):
        """self-check at start-up: RCCL's uint8 sum wraps modulo 256 (200 x world) in both collectives"""
        import torch
        w = self.world
        want = (200 * w) % 256
        src = torch.full((64 * w,), 200, dtype=torch.uint8, device=self.device)
        blk = torch.empty(64, dtype=torch.uint8, device=self.device)
        st = torch.cuda.current_stream().cuda_stream
        self.reduce_scatter_tables(src, blk, 64, st)
        self.allreduce_tables(src, st)
        torch.cuda.synchronize()
        if not bool((blk == want).all()) or not bool((src == want).all()):
            raise RuntimeError(f"RCCL uint8 sum does not wrap modulo 256: got {int(blk[0])}/{int(src[0])}, want {want}")

    def count(self):
        """the ranks RCCL counts in the communicator (ncclCommCount)"""
        import ctypes as C
        k = C.c_int(0)
        self._check(self.lib.lime_comm_count(self.h, C.byref(k)))
        return int(k.value)

    def barrier(self):
        import torch.distributed as dist
        dist.barrier(group=self.group)

    def max_float(self, x):
        import torch
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def close(self):
        if self.h:
            self.lib.lime_comm_destroy(self.h)
            self.h = None


class HostComm:
    """Rehearsal stand-in for Comm on a one-GPU box (gloo backend, every rank on GPU 0): the same calls with the
    exchange staged through host memory.  Never the measured path."""

    def __init__(self, rank, world, device, group=None):
        self.rank, self.world, self.device, self.group = rank, world, device, group

    def reduce_scatter_tables(self, sim_t, blk_t, blk_bytes, stream=None):
        import torch
        import torch.distributed as dist
        torch.cuda.synchronize()
        h = sim_t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
        blk_t.copy_(h.view(self.world, -1)[self.rank][:blk_bytes])

    def allreduce_tables(self, sim_t, stream=None):
        import torch
        import torch.distributed as dist
        torch.cuda.synchronize()
        h = sim_t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
        sim_t.copy_(h)

    def combine_counters(self, n_clusters, max_len):
        return combine_counters(n_clusters, max_len, "cpu", self.group)

    def exchange_records(self, ctx, n_reads, n_refs, block_t, stream=None):
        """the same exchange staged through host memory (gloo all_gather_object of the ranks' records)"""
        import ctypes as C
        import numpy as np
        import torch
        import torch.distributed as dist
        from . import _lib
        R, base = ctx.records_get(stream)
        total = int(base[-1])
        rec = torch.empty(max(total, 1), dtype=torch.int32, device=self.device)
        if total:
            assert _lib.hip_memcpy_d2d(rec.data_ptr(), R.d_recs, total * 4) == 0
        big = torch.empty(max(int(R.n_bigrecs), 1), dtype=torch.int64, device=self.device)
        if R.n_bigrecs:
            assert _lib.hip_memcpy_d2d(big.data_ptr(), R.d_bigrecs, int(R.n_bigrecs) * 8) == 0
        mine = (base, rec[:total].cpu().numpy(), big[:int(R.n_bigrecs)].cpu().numpy())
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=self.group)
        n_bins, bin_shift = int(R.n_bins), int(R.bin_shift)
        per = (n_bins + self.world - 1) // self.world
        b0 = min(per * self.rank, n_bins); b1 = min(b0 + per, n_bins); nb = b1 - b0
        from .api import sim_bytes as _sb
        cell_lo = b0 << bin_shift
        block_bytes = max(min(b1 << bin_shift, _sb(n_reads, n_refs)) - cell_lo, 0)
        srcoff = np.zeros((self.world, nb + 1), dtype=np.uint64)
        parts, at = [], 0
        for s, (bs, rs, _) in enumerate(everyone):
            sl = bs[b0:b1 + 1].astype(np.int64)
            srcoff[s] = at + (sl - sl[0]); parts.append(rs[int(sl[0]):int(sl[-1])]); at += int(sl[-1] - sl[0])
        rx = torch.from_numpy(np.concatenate(parts) if at else np.zeros(1, np.int32)).to(self.device)
        bigs = np.concatenate([b for _, _, b in everyone]) if sum(len(b) for _, _, b in everyone) else np.zeros(0, np.int64)
        bt = torch.from_numpy(bigs if len(bigs) else np.zeros(1, np.int64)).to(self.device)
        ctx.apply_records_dev(self.world, rx, srcoff, nb, bin_shift, bt, len(bigs), cell_lo, block_bytes, block_t, stream)
        torch.cuda.synchronize()
        return cell_lo, block_bytes

    def count(self):
        return self.world                                   # (no RCCL communicator in the rehearsal)

    def check_uint8_sum_wraps(self):
        import torch
        import torch.distributed as dist
        t = torch.full((64,), 200, dtype=torch.uint8)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        if not bool((t == (200 * self.world) % 256).all()):
            raise RuntimeError("uint8 all-reduce does not wrap modulo 256")

    def barrier(self):
        import torch.distributed as dist
        dist.barrier(group=self.group)

    def max_float(self, x):
        import torch
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def close(self):
        pass


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
    """Self-check that the backend's uint8 SUM wraps modulo 256 (200 + 100 -> 44 for 2 ranks) in the
    all-reduce; returns True if the reduce-scatter form passes the same check too (else the caller
    falls back to allreduce_tables)."""
    import torch
    import torch.distributed as dist
    w = dist.get_world_size(group)
    t = torch.full((64,), 200, dtype=torch.uint8, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    want = (200 * w) % 256
    if not bool((t == want).all()):
        raise RuntimeError(f"uint8 all-reduce does not wrap modulo 256: got {int(t[0])}, want {want}")
    ok = torch.ones(1, dtype=torch.int32, device=device)
    try:
        src = torch.full((64 * w,), 200, dtype=torch.uint8, device=device)
        blk = torch.empty(64, dtype=torch.uint8, device=device)
        reduce_scatter_tables(src, blk, group)
        if not bool((blk == want).all()):
            ok.zero_()
    except Exception:
        ok.zero_()
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)      # every rank takes the same decision
    return bool(ok.item())
