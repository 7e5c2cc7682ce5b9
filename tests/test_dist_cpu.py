"""Multi-rank logic on CPU (gloo, world_size 2): position-range sharding, the one all-reduce of the
per-rank uint8 tables (sum modulo 256) and the counter combination.  The per-shard scan itself is
stood in for by the oracle restricted to the clusters a shard owns -- the ownership rule
(a cluster belongs to the shard holding its first position) is what is under test here."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lime_amd.dist import (DEFAULT_HALO, allreduce_tables, check_uint8_sum_wraps, combine_counters,  # noqa: E402
                           reduce_scatter_tables, shard_ranges, table_block_bytes)
from lime_amd._lib import TILE  # noqa: E402


def test_shard_ranges_cover_and_align():
    for n in (1, 4095, 4096, 4097, 10**6 + 3, 10**8):
        for world in (1, 2, 3, 8):
            rs = shard_ranges(n, world)
            assert rs[0][0] == 0 and rs[-1][1] == n
            for (lo, hi, hh), nxt in zip(rs, rs[1:] + [None]):
                assert lo <= hi <= hh <= n
                assert lo % TILE == 0
                if nxt is not None:
                    assert hi == nxt[0]
                    assert hh == min(hi + DEFAULT_HALO, n) or hi == n
                else:
                    assert hh == n


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, nr, ng, out):
    from oracle import oracle_py as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        check_uint8_sum_wraps("cpu")
        lcp, da, eb = O.synth(11, 0, n, nr, ng, 16, 1)
        lcp[60000:60700] = 20                          # a run across the shard cut
        lo, hi, hh = shard_ranges(n, world, halo=8192)[rank]
        cl, _, _ = O.detect(lcp, da, nr, 16)
        mine = cl[(cl[:, 0] >= lo) & (cl[:, 0] < hi)]   # ownership: first position inside [lo, hi)
        assert (mine[:, 0] + mine[:, 1] <= hh).all(), "an owned cluster leaves the halo"
        sim = torch.from_numpy(O.score(da, eb, mine, nr, ng))
        sim += 250                                      # force wrap-around in the reduction
        # the reduce-scatter form first (bench.py): rank r ends up with block r of the summed table
        blk = table_block_bytes(sim.numel(), world)
        padded = torch.zeros(blk * world, dtype=torch.uint8)
        padded[:sim.numel()] = sim.reshape(-1)
        mine_blk = torch.empty(blk, dtype=torch.uint8)
        reduce_scatter_tables(padded.clone(), mine_blk)
        allreduce_tables(sim)
        whole = torch.zeros(blk * world, dtype=torch.uint8)
        whole[:sim.numel()] = sim.reshape(-1)
        assert torch.equal(mine_blk, whole[rank * blk:(rank + 1) * blk]), "reduce-scatter block differs"
        nc, ml = combine_counters(len(mine), int(mine[:, 1].max()) if len(mine) else 0, "cpu")
        if rank == 0:
            exp = O.score(da, eb, cl, nr, ng)
            exp = (exp.astype(np.int64) + 250 * world).astype(np.uint8)
            out.put((bool(np.array_equal(sim.numpy(), exp)), nc == len(cl), ml == int(cl[:, 1].max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_two_rank_sharding_allreduce_mod256(world):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 131072 + 77, 40, 5, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=10) == (True, True, True)


def test_choose_exchange_moves_fewer_bytes():
    """dense = the table over the links, sparse = 4 bytes per update (lime_amd/dist.py:choose_exchange); the shapes of
    BASELINE.json configs[4] over 8 GPUs and of a sparse pass over a small table"""
    from lime_amd.dist import choose_exchange
    assert choose_exchange(150_000_000, 1_000_000_000) == "sparse"      # N = 1e10 over 8 ranks: 0.6 GB of records, 1 GB table
    assert choose_exchange(3_500_000, 50_000_000) == "sparse"           # configs[1]: 14 MB of records, 50 MB table
    assert choose_exchange(120_000_000, 50_000_000) == "dense"
    assert choose_exchange(0, 16) == "sparse"

