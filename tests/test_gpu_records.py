"""Owner-partitioned exchange of table updates (include/lime_hip.h: lime_fused_records_dev / lime_records_get /
lime_apply_records_dev; transport lime_comm_exchange_records).  One process, one GPU: G position-range shards leave their
updates as records grouped by table bin; for every owner the slices of its bins are gathered from all shards (what the
all-to-all of lime_comm_exchange_records does over xGMI) and its block of the table is built from them alone.  The blocks
laid end to end must equal the table of one fused pass -- which the other tests pin to the oracle -- bit for bit."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


def _exchange_on_one_gpu(lime_amd, torch, lcp, da, eb, n, nr, ng, alpha, world, levels=None, monkeypatch=None):
    from lime_amd.dist import shard_ranges
    if levels:
        monkeypatch.setenv("LIME_BIN_LEVELS", levels)
    ctxs = [lime_amd.Context() for _ in range(world)]
    dev = torch.device("cuda:0")
    tl = torch.from_numpy(lcp.view(np.int32)).to(dev); td = torch.from_numpy(da.view(np.int32)).to(dev)
    te = None if eb is None else torch.from_numpy(eb).to(dev)
    n_bins, bin_shift = ctxs[0].records_layout(nr, ng)
    recs, bases, bigs, tot_c, tot_m, edges = [], [], [], 0, 0, []
    for r, (lo, hi, hh) in enumerate(shard_ranges(n, world)):
        c = ctxs[r]
        c.fused_records_dev(tl[lo:], td[lo:], None if te is None else te[lo:], hi - lo, hh - lo, hh == n, nr, ng, alpha)
        s, rc = c.stats(); assert rc == 0, rc
        tot_c += s.n_clusters; tot_m = max(tot_m, s.max_len); edges.append(s.edge)
        R, base = c.records_get()
        assert (R.n_bins, R.bin_shift) == (n_bins, bin_shift) and base[0] == 0
        total = int(base[-1])
        rt = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        if total:
            assert lime_amd._lib.hip_memcpy_d2d(rt.data_ptr(), R.d_recs, total * 4) == 0
        bt = torch.empty(max(int(R.n_bigrecs), 1), dtype=torch.int64, device=dev)
        if R.n_bigrecs:
            assert lime_amd._lib.hip_memcpy_d2d(bt.data_ptr(), R.d_bigrecs, int(R.n_bigrecs) * 8) == 0
        recs.append(rt); bases.append(base); bigs.append(bt[:int(R.n_bigrecs)])
    big_all = torch.cat(bigs) if sum(len(b) for b in bigs) else torch.empty(1, dtype=torch.int64, device=dev)
    n_big = sum(len(b) for b in bigs)
    sim_bytes = lime_amd.sim_bytes(nr, ng)
    per = (n_bins + world - 1) // world
    blocks = []
    for r in range(world):
        b0 = min(per * r, n_bins); b1 = min(b0 + per, n_bins); nb = b1 - b0
        cell_lo = b0 << bin_shift
        block_bytes = max(min(b1 << bin_shift, sim_bytes) - cell_lo, 0)
        srcoff = np.zeros((world, nb + 1), dtype=np.uint64)
        parts, at = [], 0
        for s in range(world):
            sl = bases[s][b0:b1 + 1].astype(np.int64)
            srcoff[s] = at + (sl - sl[0])
            parts.append(recs[s][int(sl[0]):int(sl[-1])]); at += int(sl[-1] - sl[0])
        rx = torch.cat(parts) if at else torch.empty(1, dtype=torch.int32, device=dev)
        blk = torch.full((max(block_bytes, 16),), 0xAB, dtype=torch.uint8, device=dev)     # every byte must be written
        ctxs[r].apply_records_dev(world, rx, srcoff, nb, bin_shift, big_all, n_big, cell_lo, block_bytes, blk)
        torch.cuda.synchronize()
        blocks.append(blk[:block_bytes])
    for c in ctxs:
        c.close()
    return torch.cat(blocks)[:nr * ng].cpu().numpy().reshape(nr, ng), tot_c, tot_m, edges


@pytest.mark.parametrize("n,nr,ng,mode,world,levels", [
    (300000, 1000, 50, 0, 2, None),            # one bin for the whole table: rank 1 owns nothing
    (2000003, 5000, 120, 0, 3, None),          # 600 KB: 10 regions = bins over 3 owners, the last region partial
    (3000000, 40000, 700, 1, 4, None),         # 28 MB, 214 one-region bins
    (1500000, 3000, 300, 1, 2, "2,3"),         # forced second level: 3 bins of several regions
    (1500000, 2, 131072, 0, 3, None),          # rows as long as a region
    (700001, 3, 5, 1, 2, None),                # very few documents: repeated documents everywhere, scores above 1
])
def test_blocks_built_by_their_owners_equal_the_one_pass_table(monkeypatch, n, nr, ng, mode, world, levels):
    import torch
    import lime_amd
    from lime_amd.dist import combine_edges
    lcp, da, eb = O.synth(3000 + n, 0, n, nr, ng, 16, mode)
    lcp[n // 2:n // 2 + 700] = 20              # a run of 700 across a shard cut: a long cluster (if it holds a read and a genome)
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    for e in (eb, None):
        exp = O.score(da, e, cl, nr, ng, threads=4)
        got, tot_c, tot_m, edges = _exchange_on_one_gpu(lime_amd, torch, lcp, da, e, n, nr, ng, 16, world, levels, monkeypatch)
        combine_edges(edges)
        assert (tot_c, tot_m) == (nc, ml)
        assert np.array_equal(got, exp)


def test_records_of_a_pool_that_is_too_small_are_repaired(monkeypatch):
    """LIME_POOL_DENSITY far below the real update density: the first pass overflows, lime_get_stats repeats it"""
    import torch
    import lime_amd
    monkeypatch.setenv("LIME_POOL_DENSITY", "0.0005")
    monkeypatch.setenv("LIME_POOL_SLACK", "0")
    n, nr, ng = 2000000, 2000, 64
    lcp, da, eb = O.synth(77, 0, n, nr, ng, 16, 1)
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    exp = O.score(da, eb, cl, nr, ng, threads=4)
    got, tot_c, tot_m, _ = _exchange_on_one_gpu(lime_amd, torch, lcp, da, eb, n, nr, ng, 16, 2)
    assert (tot_c, tot_m) == (nc, ml) and np.array_equal(got, exp)


def test_blocks_at_the_shape_of_configs3():
    """BASELINE.json configs[3]'s shapes (setB2: 20 249 373 reads x 930 genomes = 18.8 GB table, README.md:137; 4 GPUs) on
    2 * 10^9 synthetic symbols generated on the device, EBWT=1: four position-range shards leave their updates as records (five
    sub-regions per wave: the table is beyond 16 GB), every owner gathers the slices of its quarter of the bins and builds
    its 4.7 GB block alone; block r must equal bytes [cell_lo, cell_lo + block_bytes) of the table of ONE fused pass over the
    whole collection (which tests/test_gpu_parity.py::test_full_size_paths_agree[C4] ties to the other paths)."""
    import torch
    import lime_amd
    from lime_amd.dist import shard_ranges, combine_edges
    n, nr, ng, alpha, world = 2_000_000_000, 20_249_373, 930, 16, 4
    dev = torch.device("cuda:0")
    lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
    eb = torch.empty(n, dtype=torch.uint8, device=dev)
    c0 = lime_amd.Context()
    c0.synth_dev(42, 0, n, nr, ng, alpha, 0, lcp, da, eb)
    tb = lime_amd.sim_bytes(nr, ng)
    A = torch.empty(tb, dtype=torch.uint8, device=dev)
    c0.fused_dev(lcp, da, eb, n, n, True, nr, ng, alpha, A, True)
    sA, rc = c0.stats(); assert rc == 0
    n_bins, bin_shift = c0.records_layout(nr, ng)
    c0.close()
    ctxs = [lime_amd.Context() for _ in range(world)]
    recs, bases, bigs, tot_c, tot_u, edges = [], [], [], 0, 0, []
    for r, (lo, hi, hh) in enumerate(shard_ranges(n, world)):
        c = ctxs[r]
        c.fused_records_dev(lcp[lo:], da[lo:], eb[lo:], hi - lo, hh - lo, hh == n, nr, ng, alpha)
        s, rc = c.stats(); assert rc == 0, rc
        tot_c += s.n_clusters; tot_u += s.n_updates; edges.append(s.edge)
        R, base = c.records_get()
        assert (R.n_bins, R.bin_shift) == (n_bins, bin_shift)
        recs.append(R); bases.append(base)
        bt = torch.empty(max(int(R.n_bigrecs), 1), dtype=torch.int64, device=dev)      # updates of clusters that left the scan (a few window-crossing ones)
        if R.n_bigrecs:
            assert lime_amd._lib.hip_memcpy_d2d(bt.data_ptr(), R.d_bigrecs, int(R.n_bigrecs) * 8) == 0
        bigs.append(bt[:int(R.n_bigrecs)])
    combine_edges(edges)
    assert (tot_c, tot_u) == (sA.n_clusters, sA.n_updates)
    del lcp, da, eb
    per = (n_bins + world - 1) // world
    n_big = sum(len(b) for b in bigs)
    big_all = torch.cat(bigs) if n_big else torch.empty(1, dtype=torch.int64, device=dev)
    for r in range(world):
        b0 = min(per * r, n_bins); b1 = min(b0 + per, n_bins); nb = b1 - b0
        cell_lo = b0 << bin_shift
        block_bytes = max(min(b1 << bin_shift, tb) - cell_lo, 0)
        srcoff = np.zeros((world, nb + 1), dtype=np.uint64)
        total = sum(int(bases[s][b1] - bases[s][b0]) for s in range(world))
        rx = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        at = 0
        for s in range(world):
            sl = bases[s][b0:b1 + 1].astype(np.int64)
            srcoff[s] = at + (sl - sl[0])
            cnt = int(sl[-1] - sl[0])
            if cnt:
                assert lime_amd._lib.hip_memcpy_d2d(rx.data_ptr() + 4 * at, recs[s].d_recs + 4 * int(sl[0]), 4 * cnt) == 0
            at += cnt
        blk = torch.full((max(block_bytes, 16),), 0xAB, dtype=torch.uint8, device=dev)     # every byte must be written
        ctxs[r].apply_records_dev(world, rx, srcoff, nb, bin_shift, big_all, n_big, cell_lo, block_bytes, blk)
        torch.cuda.synchronize()
        assert torch.equal(blk[:block_bytes], A[cell_lo:cell_lo + block_bytes]), r
        del blk, rx
    for c in ctxs:
        c.close()

