"""The binned table-update path (records -> bins -> table regions built in LDS) against the
oracle and the reference's golden vectors, bit-exact, through the C ABI.  LIME_UPDATE_PATH=bin forces the path for
inputs of any size (by default it is chosen for update-dense passes over large tables only)."""
import numpy as np
import pytest

from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bctx():
    import os
    import lime_amd
    old = os.environ.get("LIME_UPDATE_PATH")
    os.environ["LIME_UPDATE_PATH"] = "bin"
    c = lime_amd.Context()
    if old is None:
        del os.environ["LIME_UPDATE_PATH"]
    else:
        os.environ["LIME_UPDATE_PATH"] = old
    yield c
    c.close()


@pytest.mark.parametrize("ebwt_mode", [1, 0])
def test_binned_fused_golden(bctx, golden, ebwt_mode):
    eb = golden["ebwt"] if ebwt_mode else None
    sim, nc, ml = bctx.fused(golden["lcp"], golden["da"], eb, golden["n_reads"], golden["n_refs"], golden["alpha"])
    assert nc == len(golden["clrs"])
    assert np.array_equal(sim, golden[f"sim_e{ebwt_mode}"])


@pytest.mark.parametrize("n,nr,ng,mode", [
    (1, 3, 2, 0), (63, 3, 2, 0), (4097, 5, 4, 0), (12289, 40, 7, 1), (300000, 1000, 50, 0), (300001, 200, 9, 1),
    (2000003, 5000, 120, 0),          # 600 KB table: 5 regions, the last one partial
    (1500000, 1, 1, 1),               # one cell: every update collides (LDS compare-and-swap retries), wraps many times
    (1500000, 2, 131072, 0),          # rows as long as a region
    (3000000, 40000, 700, 1),         # 28 MB table, 214 regions
])
def test_binned_fused_vs_oracle(bctx, n, nr, ng, mode):
    lcp, da, eb = O.synth(2000 + n, 0, n, nr, ng, 16, mode)
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    for e in (eb, None):
        exp = O.score(da, e, cl, nr, ng, threads=4)
        sim, gnc, gml = bctx.fused(lcp, da, e, nr, ng, 16)
        s, rc = bctx.stats()
        assert rc == 0 and s.wave_records_max > 0 or nc == 0
        assert (gnc, gml) == (nc, ml)
        assert np.array_equal(sim, exp)


@pytest.mark.parametrize("second", ["tiles", "sweeps"])
@pytest.mark.parametrize("levels,nr,ng", [("1,1", 3000, 300), ("2,3", 5000, 1200), ("4,7", 40000, 700), ("1,2", 1, 1)])
def test_binned_two_levels_on_small_tables(monkeypatch, levels, nr, ng, second):
    """LIME_BIN_LEVELS forces bins of several regions (second partition level) on tables of a few regions: 1 bin for
    the whole table, 3 bins of 32 regions, ... ; one cell only.  Second level: tile by tile (k_sort_tiles +
    k_apply_tiles, the default) and the two-sweep kernels kept for comparison runs (k_part2 + k_apply)."""
    import lime_amd
    monkeypatch.setenv("LIME_UPDATE_PATH", "bin")
    monkeypatch.setenv("LIME_BIN_LEVELS", levels)
    monkeypatch.setenv("LIME_SECOND_LEVEL", second)
    c = lime_amd.Context()
    try:
        n = 1200000
        lcp, da, eb = O.synth(500 + nr, 0, n, nr, ng, 16, 1)
        cl, nc, ml = O.detect(lcp, da, nr, 16)
        for e in (eb, None):
            exp = O.score(da, e, cl, nr, ng, threads=4)
            sim, gnc, gml = c.fused(lcp, da, e, nr, ng, 16)
            assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp)
    finally:
        c.close()


def test_binned_bins_wider_than_a_region(bctx):
    """a table of more than 1024 regions of 64 KB: 430 bins of 16 regions, second level"""
    n, nr, ng = 3000000, 150000, 3000                      # 450 MB
    lcp, da, _ = O.synth(31, 0, n, nr, ng, 16, 0)
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    exp = O.score(da, None, cl, nr, ng, threads=4)
    sim, gnc, gml = bctx.fused(lcp, da, None, nr, ng, 16)
    assert (gnc, gml) == (nc, ml)
    assert np.array_equal(sim, exp)


def test_binned_with_long_clusters(bctx):
    """clusters longer than 64 go to k_score_big, whose compare-and-swaps must land on the table k_apply built"""
    rng = np.random.default_rng(11)
    n = 200000
    lcp = np.where(rng.random(n) < 0.97, 20, 3).astype(np.uint32)
    lcp[0] = 0
    lcp[20000:45000] = 30
    lcp[45000] = 1
    da = np.where(rng.random(n) < 0.5, rng.integers(0, 3, n), 3 + rng.integers(0, 4, n)).astype(np.uint32)
    eb = rng.choice(np.frombuffer(b"ACGTNRY\x00", np.uint8), n).astype(np.uint8)
    cl, nc, ml = O.detect(lcp, da, 3, 16)
    for e in (eb, None):
        exp = O.score(da, e, cl, 3, 4, threads=4)
        sim, gnc, gml = bctx.fused(lcp, da, e, 3, 4, 16)
        assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp)


def test_binned_pool_too_small_is_repeated(monkeypatch):
    """a record pool sized for 1 update per 1000 symbols: the pass overflows it, lime_get_stats repeats the pass with
    a pool sized from what the first attempt counted, and the result is the oracle's"""
    import lime_amd
    monkeypatch.setenv("LIME_UPDATE_PATH", "bin")
    monkeypatch.setenv("LIME_POOL_DENSITY", "0.001")
    monkeypatch.setenv("LIME_POOL_SLACK", "0")
    c = lime_amd.Context()
    try:
        n, nr, ng = 2500000, 3000, 300
        lcp, da, eb = O.synth(77, 0, n, nr, ng, 16, 1)
        cl, nc, ml = O.detect(lcp, da, nr, 16)
        for e in (None, eb):
            exp = O.score(da, e, cl, nr, ng, threads=4)
            sim, gnc, gml = c.fused(lcp, da, e, nr, ng, 16)
            assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp)
    finally:
        c.close()


@pytest.mark.parametrize("n_shards", [2, 3])
def test_binned_shards_sum_to_whole(bctx, n_shards):
    import torch
    import lime_amd
    from lime_amd.dist import shard_ranges
    n, nr, ng = 700001, 200, 1300
    lcp, da, eb = O.synth(9, 0, n, nr, ng, 16, 1)
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    exp = O.score(da, eb, cl, nr, ng, threads=4)
    total = np.zeros((nr, ng), np.uint8)
    tot_c = 0
    for lo, hi, hi_halo in shard_ranges(n, n_shards, halo=65536 + 4096):
        tl = torch.from_numpy(lcp[lo:hi_halo].view(np.int32)).cuda()
        td = torch.from_numpy(da[lo:hi_halo].view(np.int32)).cuda()
        te = torch.from_numpy(eb[lo:hi_halo]).cuda()
        sim = torch.full((lime_amd.sim_bytes(nr, ng),), 7, dtype=torch.uint8, device="cuda")   # not cleared: the path writes every byte
        bctx.fused_dev(tl, td, te, hi - lo, hi_halo - lo, hi_halo == n, nr, ng, 16, sim)
        s, rc = bctx.stats()
        assert rc == 0
        tot_c += s.n_clusters
        total = (total + sim[:nr * ng].cpu().numpy().reshape(nr, ng)).astype(np.uint8)
    assert tot_c == nc
    assert np.array_equal(total, exp)


@pytest.mark.parametrize("seed", range(12))
def test_binned_random_shapes(monkeypatch, seed):
    """random table shapes around region and bin edges (rows that are not multiples of 4 bytes, tables of exactly k
    regions +- a few bytes, one row, one column), random level limits, both builds, against the oracle"""
    import lime_amd
    rng = np.random.default_rng(1000 + seed)
    kind = seed % 4
    if kind == 0:                                   # table ends a few bytes around a multiple of 64 KB
        ng = int(rng.integers(3, 900)); nr = max(1, (int(rng.integers(1, 6)) * 65536 + int(rng.integers(-5, 6))) // ng)
    elif kind == 1:                                 # one read, many genomes / one genome, many reads
        nr, ng = (1, int(rng.integers(70000, 200000))) if seed % 8 == 1 else (int(rng.integers(70000, 200000)), 1)
    elif kind == 2:                                 # odd widths
        nr, ng = int(rng.integers(50, 3000)), int(rng.integers(1, 40)) * 2 + 1
    else:
        nr, ng = int(rng.integers(1000, 20000)), int(rng.integers(100, 2000))
    levels = f"{int(rng.integers(1, 40))},{int(rng.integers(1, 300))}"
    monkeypatch.setenv("LIME_UPDATE_PATH", "bin")
    monkeypatch.setenv("LIME_BIN_LEVELS", levels)
    n = int(rng.integers(200000, 900000))
    lcp, da, eb = O.synth(7000 + seed, 0, n, nr, ng, 16, seed & 1)
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    c = lime_amd.Context()
    try:
        for e in (eb, None):
            exp = O.score(da, e, cl, nr, ng, threads=4)
            sim, gnc, gml = c.fused(lcp, da, e, nr, ng, 16)
            assert (gnc, gml) == (nc, ml), (nr, ng, levels)
            assert np.array_equal(sim, exp), (nr, ng, levels, int((sim != exp).sum()))
    finally:
        c.close()


@pytest.mark.parametrize("nr,ng,levels", [(3000, 300, None), (200000, 3000, None), (150000, 3000, "4,7")])
def test_binned_pool_overflows_badly_and_is_repeated(monkeypatch, nr, ng, levels):
    """a pool with room for a few records per wave on an input with hundreds per window (LIME_POOL_SLACK=0): nearly every
    record of the first attempt is dropped -- its counters and lists must stay consistent with what WAS stored (nothing may be
    written past the record buffers) -- and lime_get_stats repeats the pass until the table is the oracle's"""
    import lime_amd
    monkeypatch.setenv("LIME_UPDATE_PATH", "bin")
    monkeypatch.setenv("LIME_POOL_DENSITY", "0.0001")
    monkeypatch.setenv("LIME_POOL_SLACK", "0")
    if levels:
        monkeypatch.setenv("LIME_BIN_LEVELS", levels)
    c = lime_amd.Context()
    try:
        n = 6000000
        lcp, da, eb = O.synth(91, 0, n, nr, ng, 16, 1)
        cl, nc, ml = O.detect(lcp, da, nr, 16)
        for e in (eb, None):
            exp = O.score(da, e, cl, nr, ng, threads=4)
            sim, gnc, gml = c.fused(lcp, da, e, nr, ng, 16)
            assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp)
    finally:
        c.close()


def test_overflowed_pass_followed_by_another_without_stats_is_an_error(monkeypatch):
    """Two different shards through one ctx with no lime_get_stats between them, the FIRST denser than the pool: its table is
    short and can no longer be repaired (the arrays were replaced).  The next lime_get_stats must say so (LIME_ERR_NOMEM)
    instead of returning a quietly incomplete table; the ctx is usable again afterwards."""
    import torch
    import lime_amd
    from lime_amd._lib import ERR_NOMEM
    monkeypatch.setenv("LIME_UPDATE_PATH", "bin")
    monkeypatch.setenv("LIME_POOL_DENSITY", "0.001")
    monkeypatch.setenv("LIME_POOL_SLACK", "0")
    c = lime_amd.Context()
    try:
        n, nr, ng = 1500000, 3000, 300
        dense = O.synth(5, 0, n, nr, ng, 16, 1)                       # ~0.1 updates per symbol: overflows a pool for 0.001
        lcp2 = np.zeros(n, np.uint32); da2 = np.zeros(n, np.uint32)   # no clusters at all: fits any pool
        dev = "cuda"
        t1 = [torch.from_numpy(x.view(np.int32) if x.dtype == np.uint32 else x).to(dev) for x in dense]
        t2 = [torch.from_numpy(lcp2.view(np.int32)).to(dev), torch.from_numpy(da2.view(np.int32)).to(dev)]
        sim1 = torch.empty(lime_amd.sim_bytes(nr, ng), dtype=torch.uint8, device=dev)
        sim2 = torch.empty_like(sim1)
        c.fused_dev(t1[0], t1[1], t1[2], n, n, True, nr, ng, 16, sim1)      # overflows; not settled
        c.fused_dev(t2[0], t2[1], None, n, n, True, nr, ng, 16, sim2)       # another pass on the ctx
        s, rc = c.stats()
        assert rc == ERR_NOMEM
        # settled one by one, both are right
        cl, nc, ml = O.detect(dense[0], dense[1], nr, 16)
        exp = O.score(dense[1], dense[2], cl, nr, ng, threads=4)
        c.fused_dev(t1[0], t1[1], t1[2], n, n, True, nr, ng, 16, sim1)
        s, rc = c.stats(); assert rc == 0 and s.n_clusters == nc
        assert np.array_equal(sim1[:nr * ng].cpu().numpy().reshape(nr, ng), exp)
        c.fused_dev(t2[0], t2[1], None, n, n, True, nr, ng, 16, sim2)
        s, rc = c.stats(); assert rc == 0 and s.n_clusters == 0 and int(sim2.count_nonzero()) == 0
    finally:
        c.close()


def test_stream_chunks_stay_off_the_binned_path(monkeypatch):
    """lime_fused_stream with the binned path forced, a pool far too small and several chunks: chunk 0 must not leave
    records behind that a later repeat would apply to buffers holding another chunk (round 2 advisor finding)"""
    import lime_amd
    monkeypatch.setenv("LIME_UPDATE_PATH", "bin")
    monkeypatch.setenv("LIME_POOL_DENSITY", "0.001")
    monkeypatch.setenv("LIME_POOL_SLACK", "0")
    c = lime_amd.Context()
    try:
        n, nr, ng = 1300001, 2000, 200
        lcp, da, eb = O.synth(41, 0, n, nr, ng, 16, 1)
        cl, nc, ml = O.detect(lcp, da, nr, 16)
        for e in (eb, None):
            exp = O.score(da, e, cl, nr, ng, threads=4)
            sim, gnc, gml = c.fused_stream(lcp, da, e, nr, ng, 16, chunk=400000)      # 4 chunks
            assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp)
            sim, gnc, gml = c.fused_stream(lcp, da, e, nr, ng, 16, chunk=2 * n)       # one chunk: binned, overflow, repeated
            assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp)
    finally:
        c.close()


# ---- the list flow (lime_score_dev: clusters from a .clrs list, arrays resident) on the binned path (round 4) -------------
def _score_dev(ctx, da, eb, cl, nr, ng, poison=0xA5):
    import torch
    import lime_amd
    dev = torch.device("cuda", 0)
    td = torch.from_numpy(da.view(np.int32)).to(dev)
    te = None if eb is None else torch.from_numpy(eb).to(dev)
    tc = torch.from_numpy(np.ascontiguousarray(cl).view(np.int64).reshape(-1)).to(dev) if len(cl) else torch.zeros(2, dtype=torch.int64, device=dev)
    sim = torch.full((lime_amd.sim_bytes(nr, ng),), poison, dtype=torch.uint8, device=dev)     # every byte must be written
    ctx.score_dev(td, te, len(da), tc.data_ptr(), len(cl), nr, ng, sim, True)
    s, rc = ctx.stats()
    torch.cuda.synchronize()
    return sim[:nr * ng].cpu().numpy().reshape(nr, ng), s, rc


@pytest.mark.parametrize("n,nr,ng,mode", [
    (300001, 200, 9, 1),
    (2000003, 5000, 120, 0),          # 600 KB table (below the 1 MB limit of the binned path: compare-and-swap, same result)
    (2500000, 30000, 90, 1),          # 2.7 MB table, one level
    (3000000, 40000, 700, 1),         # 28 MB table, 214 regions
])
def test_list_scoring_on_the_binned_path_vs_oracle(bctx, n, nr, ng, mode):
    """ClusterBWT_DA.cpp:301-340 with the arrays on the device: k_score_list emits records, k_part -> k_apply build the table,
    k_score_big adds the long clusters; the clusters in the list's (shuffled) order"""
    lcp, da, eb = O.synth(4000 + n, 0, n, nr, ng, 16, mode)
    lcp[n // 2:n // 2 + 900] = 40                              # a long cluster for k_score_big
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    rng = np.random.default_rng(5)
    cls = cl[rng.permutation(len(cl))]
    for e in (eb, None):
        exp = O.score(da, e, cl, nr, ng, threads=4)
        got, s, rc = _score_dev(bctx, da, e, cls, nr, ng)
        assert rc == 0
        assert np.array_equal(got, exp)
        assert (s.wave_records_max > 0) == (nr * ng >= (1 << 20)), "the binned path is taken from 1 MB of table on"



def test_list_scoring_binned_with_second_level_and_small_pool(monkeypatch):
    """forced second level (bins of several regions) and a pool far too small: the pass is found incomplete inside the call and the list
    scored again by compare-and-swap -- same table"""
    import lime_amd
    n, nr, ng = 1500000, 40000, 700
    lcp, da, eb = O.synth(77, 0, n, nr, ng, 16, 1)
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    exp = O.score(da, eb, cl, nr, ng, threads=4)
    monkeypatch.setenv("LIME_UPDATE_PATH", "bin")
    monkeypatch.setenv("LIME_BIN_LEVELS", "4,7")
    c = lime_amd.Context()
    try:
        got, s, rc = _score_dev(c, da, eb, cl, nr, ng)
        assert rc == 0 and np.array_equal(got, exp)
    finally:
        c.close()
    monkeypatch.delenv("LIME_BIN_LEVELS")
    monkeypatch.setenv("LIME_POOL_DENSITY", "0.0005"); monkeypatch.setenv("LIME_POOL_SLACK", "0")
    c = lime_amd.Context()
    try:
        got, s, rc = _score_dev(c, da, eb, cl, nr, ng)
        assert rc == 0 and np.array_equal(got, exp)
    finally:
        c.close()


def test_whole_line_partition_on_a_table_of_two_sub_regions(monkeypatch):
    """k_part_lines (bins few enough for their line buffers: forced here, 298 bins of 256 regions) on a 5 GB table: the cells beyond 4 GB
    come from the waves' second sub-regions (bin = record >> shift + the sub-region's offset); binned table == compare-and-swap table"""
    import torch
    import lime_amd
    n, nr, ng = 30_000_000, 1_000_000, 5000
    dev = torch.device("cuda", 0)
    lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
    tabs = []
    for path, levels in (("cas", None), ("bin", "1024,512")):
        monkeypatch.setenv("LIME_UPDATE_PATH", path)
        if levels: monkeypatch.setenv("LIME_BIN_LEVELS", levels)
        c = lime_amd.Context()
        try:
            c.synth_dev(9, 0, n, nr, ng, 16, 0, lcp, da, None)
            sim = torch.empty(lime_amd.sim_bytes(nr, ng), dtype=torch.uint8, device=dev)
            c.fused_dev(lcp, da, None, n, n, True, nr, ng, 16, sim, True)
            s, rc = c.stats()
            assert rc == 0 and (s.wave_records_max > 0) == (path == "bin")
            tabs.append((sim, int(s.n_updates)))
        finally:
            c.close()
    assert tabs[0][1] == tabs[1][1] and torch.equal(tabs[0][0], tabs[1][0])


@pytest.mark.parametrize("levels,nr,ng,mode", [("1,1", 3000, 300, 1), ("2,3", 5000, 1200, 0), ("4,7", 40000, 700, 1), ("1,2", 1, 1, 0), (None, 150000, 3000, 0)])
def test_apply_variant_for_many_records_on_small_inputs(monkeypatch, levels, nr, ng, mode):
    """k_apply_tiles<true> (chosen at the launch from about 2e8 records on: a step's groups 64 .. 79 of four runs added in one
    pass, longer runs in a loop) forced on small inputs with LIME_APPLY_WIDE=1: one bin for the whole table (runs of
    thousands of records per tile and region: the loop), bins of a few regions (runs around 256: the shared pass), the
    default layout; wrap-around cells (few reads) take the exact second pass"""
    import lime_amd
    monkeypatch.setenv("LIME_UPDATE_PATH", "bin")
    monkeypatch.setenv("LIME_APPLY_WIDE", "1")
    if levels:
        monkeypatch.setenv("LIME_BIN_LEVELS", levels)
    c = lime_amd.Context()
    try:
        n = 1500000
        lcp, da, eb = O.synth(900 + nr, 0, n, nr, ng, 16, mode)
        cl, nc, ml = O.detect(lcp, da, nr, 16)
        for e in (eb, None):
            exp = O.score(da, e, cl, nr, ng, threads=4)
            sim, gnc, gml = c.fused(lcp, da, e, nr, ng, 16)
            assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp), (levels, nr, ng, int((sim != exp).sum()))
    finally:
        c.close()


@pytest.mark.parametrize("wide", ["0", "1"])
def test_one_bin_of_more_than_512_tiles(monkeypatch, wide):
    """the whole table one bin (LIME_BIN_LEVELS=1,1) with more records than 512 second-level tiles hold: k_apply_tiles walks the
    bin's tiles in rounds of 512 (a lane per tile and wave), reloading its index entries per round -- both variants of the kernel"""
    import lime_amd
    monkeypatch.setenv("LIME_UPDATE_PATH", "bin")
    monkeypatch.setenv("LIME_BIN_LEVELS", "1,1")
    monkeypatch.setenv("LIME_APPLY_WIDE", wide)
    c = lime_amd.Context()
    try:
        n, nr, ng = 40_000_000, 3000, 300
        lcp, da, _ = O.synth(4242, 0, n, nr, ng, 16, 0)
        cl, nc, ml = O.detect(lcp, da, nr, 16)
        exp = O.score(da, None, cl, nr, ng, threads=8)
        sim, gnc, gml = c.fused(lcp, da, None, nr, ng, 16)
        s, rc = c.stats()
        assert rc == 0 and s.n_updates > 512 * 8192, int(s.n_updates)      # more than 512 tiles in the one bin
        assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp), int((sim != exp).sum())
    finally:
        c.close()


@pytest.mark.parametrize("no_direct", ["0", "1"])
@pytest.mark.parametrize("levels", [None, "1,2", "4,7"])
def test_records_written_by_the_scorers_or_through_the_queue(monkeypatch, golden, no_direct, levels):
    """Round 6: for tables of one or two sub-regions the scan's scorers write finished 4-byte records where their 64-byte lines are gathered
    (k_scan<., 0, 2>); option no_direct keeps the update queue and its drains (k_scan<., 0, 1>, which tables of three and more sub-regions always
    take).  Both on every golden vector of the reference, both builds, three bin layouts; plus a table of two sub-regions (4.4 GB) whose rows
    straddle the 2^32 border, against the ORACLE's table compared on the device."""
    import torch
    import lime_amd
    monkeypatch.setenv("LIME_UPDATE_PATH", "bin")
    if levels:
        monkeypatch.setenv("LIME_BIN_LEVELS", levels)
    c = lime_amd.Context()
    try:
        c.set_option("no_direct", no_direct)
        g = golden
        for ebwt_on, key in ((True, "sim_e1"), (False, "sim_e0")):
            sim, nc, ml = c.fused(g["lcp"], g["da"], g["ebwt"] if ebwt_on else None, g["n_reads"], g["n_refs"], g["alpha"])
            s, rc = c.stats()
            assert rc == 0 and nc == len(g["clrs"]) and np.array_equal(sim, g[key]), (g["name"], no_direct, levels, ebwt_on)
        if g["name"] != "edges" or levels:
            return
        n, nr, ng = 3_000_000, 1_100_000, 4000                      # 4.4 GB: two sub-regions; reads around 2^32 / 4000 = 1 073 741 straddle the border
        lcp, da, eb = O.synth(99, 0, n, nr, ng, 16, 1)
        lo_r = (1 << 32) // ng - 3
        sel = da < nr
        da[sel] = (lo_r + da[sel] % 7).astype(np.uint32)            # every read id in the seven rows around the border
        cl, nc, ml = O.detect(lcp, da, nr, 16)
        dev = torch.device("cuda", 0)
        tl = torch.from_numpy(lcp.view(np.int32)).to(dev); td = torch.from_numpy(da.view(np.int32)).to(dev); te = torch.from_numpy(eb).to(dev)
        sim = torch.empty(lime_amd.sim_bytes(nr, ng), dtype=torch.uint8, device=dev)
        for e, et in ((eb, te), (None, None)):
            rows = O.score(np.where(da < nr, da - lo_r, da - nr + 7).astype(np.uint32), e, cl, 7, ng, threads=8)     # the same clusters with the 7 reads renumbered 0..6
            c.fused_dev(tl, td, et, n, n, True, nr, ng, 16, sim)
            s, rc = c.stats()
            assert rc == 0 and (s.n_clusters, s.max_len) == (nc, ml) and s.wave_records_max > 0
            got = sim[lo_r * ng:(lo_r + 7) * ng].cpu().numpy().reshape(7, ng)
            assert np.array_equal(got, rows), (no_direct, e is None, int((got != rows).sum()))
            assert int(torch.count_nonzero(sim[:lo_r * ng])) == 0 and int(torch.count_nonzero(sim[(lo_r + 7) * ng:nr * ng])) == 0
    finally:
        c.close()
