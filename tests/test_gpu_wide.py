"""Round 5: the binned update path beyond its old limits -- 64-bit positions of the binned records (passes whose
record pool holds 2^32 records or more: N = 10^10 at the update density of real text), sub-regions sized for their share of a wave's
records (tables beyond 4 GB), the sampled density probe in front of a context's first pass, and the loud compare-and-swap fallback.
All through the C ABI, bit-exact against the oracle (or, at full size, against the compare-and-swap path of the same library).
Reference site of the updates carried: ClusterBWT_DA.cpp:178-184, 243-248."""
import numpy as np
import pytest

from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


def _ctx(monkeypatch, **env):
    import lime_amd
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    return lime_amd.Context()


# (levels, nr, ng): one bin per region -> k_part + k_apply; bins of several regions, few of them -> k_part_lines + tiles;
# many bins -> k_part + tiles
P64_SHAPES = [(None, 3000, 300), ("2,3", 5000, 1200), ("4,7", 40000, 700), ("1,1", 3000, 300), (None, 150000, 3000), ("1,2", 1, 1), ("2000,2000", 60000, 2000)]


@pytest.mark.parametrize("base", [0, (1 << 32) - 4096, (1 << 32) - 300_000, (3 << 32) - 160_000, (1 << 33) + 48])
@pytest.mark.parametrize("levels,nr,ng", P64_SHAPES)
def test_p64_partition_kernels_vs_oracle(monkeypatch, base, levels, nr, ng):
    """k_part<.., true> / k_part_lines<true> (positions of the binned records as 64-bit numbers) on small passes whose record positions are
    made to START at `base` (LIME_P64_TEST_BASE: the bin bases are shifted and the kernels get the array's address minus the base), so that the
    positions cross a multiple of 2^32 inside a bin, inside a tile, between two bins -- or lie wholly above one.  base 0: LIME_FORCE_P64 alone."""
    env = {"LIME_UPDATE_PATH": "bin", "LIME_FORCE_P64": 1}
    if base:
        env["LIME_P64_TEST_BASE"] = base
    if levels:
        env["LIME_BIN_LEVELS"] = levels
    c = _ctx(monkeypatch, **env)
    try:
        n = 1_500_000
        lcp, da, eb = O.synth(300 + nr, 0, n, nr, ng, 16, 1)
        cl, nc, ml = O.detect(lcp, da, nr, 16)
        for e in (eb, None):
            exp = O.score(da, e, cl, nr, ng, threads=4)
            sim, gnc, gml = c.fused(lcp, da, e, nr, ng, 16)
            s, rc = c.stats()
            assert rc == 0 and (s.wave_records_max > 0 or nc == 0)
            assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp), (base, levels, nr, ng, int((sim != exp).sum()))
    finally:
        c.close()


def test_p64_with_a_pool_too_small(monkeypatch):
    """the repeated pass (pool sized from what the first attempt counted) on the 64-bit kernels"""
    c = _ctx(monkeypatch, LIME_UPDATE_PATH="bin", LIME_P64_TEST_BASE=(1 << 32) - 100_000, LIME_POOL_DENSITY="0.001", LIME_POOL_SLACK=0)
    try:
        n, nr, ng = 2_500_000, 3000, 300
        lcp, da, eb = O.synth(77, 0, n, nr, ng, 16, 1)
        cl, nc, ml = O.detect(lcp, da, nr, 16)
        exp = O.score(da, eb, cl, nr, ng, threads=4)
        sim, gnc, gml = c.fused(lcp, da, eb, nr, ng, 16)
        assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp)
        assert c.host_times()["repeats"] >= 1
    finally:
        c.close()


@pytest.mark.parametrize("nr,ng,ebwt_on,mode", [
    (1_100_000, 4000, True, 1),        # 4.4 GB: two sub-regions (drain_lines), the second one a tenth of the first
    (3_000_000, 3423, True, 1),        # configs[4]'s table, 10.3 GB: three sub-regions, shares 0.42 / 0.42 / 0.16
    (2_200_000, 4000, False, 1),       # 8.8 GB, EBWT=0: three sub-regions
    (4_700_000, 4000, True, 1),        # the size of configs[3]'s table, 18.8 GB: five sub-regions
])
@pytest.mark.parametrize("no_direct", [0, 1])
def test_tables_of_several_sub_regions_vs_oracle(monkeypatch, nr, ng, ebwt_on, mode, no_direct):
    """Tables beyond 4 GB: a scan wave writes its records to one sub-region per 4 GB of table, each sized for its SHARE of the wave's
    records (round 5; rounds 3-4: the wave's whole share each) -- clustered generator, 2*10^7 symbols, against the oracle's table (compared on
    the device).  The density comes from the probe; no pass may be repeated and none may fall back.  Records written by the scorers into a
    line buffer per sub-region (k_scan<., 0, 2> and <., 0, 3>, round 6) and, with option no_direct, through the update queue (drain_lines /
    drain_bin of k_scan<., 0, 1> -- the path of tables beyond 24 / 32 GB)."""
    import torch
    c = _ctx(monkeypatch, LIME_UPDATE_PATH="bin", LIME_PROBE_MIN=1 << 24, LIME_NO_DIRECT=no_direct)      # (the probe in front of first passes of 2^28 symbols and more by default)
    try:
        n = 20_000_000
        lcp, da, eb = O.synth(4100 + ng, 0, n, nr, ng, 16, mode)
        e = eb if ebwt_on else None
        cl, nc, ml = O.detect(lcp, da, nr, 16)
        exp = O.score(da, e, cl, nr, ng, threads=8)
        dev = torch.device("cuda", 0)
        import lime_amd
        tl = torch.from_numpy(lcp.view(np.int32)).to(dev); td = torch.from_numpy(da.view(np.int32)).to(dev)
        te = torch.from_numpy(eb).to(dev) if ebwt_on else None
        sim = torch.full((lime_amd.sim_bytes(nr, ng),), 0x5A, dtype=torch.uint8, device=dev)      # every byte must be written
        c.fused_dev(tl, td, te, n, n, True, nr, ng, 16, sim, True)
        s, rc = c.stats()
        assert rc == 0 and (s.n_clusters, s.max_len) == (nc, ml) and s.wave_records_max > 0
        ht = c.host_times()
        assert ht["probes"] == 1 and ht["repeats"] == 0 and ht["cas_fallbacks"] == 0, ht
        assert abs(ht["records_per_symbol"] - s.n_updates / n) < 1e-9                # (after the pass: what it counted)
        texp = torch.from_numpy(exp.reshape(-1)).to(dev)
        assert torch.equal(sim[:nr * ng], texp), int((sim[:nr * ng] != texp).sum())
        del texp
        # the same pass with the sub-regions far too small: overflow, repeat sized from the fullest SUB-REGION's count
        c2 = _ctx(monkeypatch, LIME_UPDATE_PATH="bin", LIME_POOL_DENSITY="0.002", LIME_POOL_SLACK=0, LIME_NO_DIRECT=no_direct)
        try:
            sim2 = torch.full_like(sim, 0xA5)
            c2.fused_dev(tl, td, te, n, n, True, nr, ng, 16, sim2, True)
            s2, rc2 = c2.stats()
            assert rc2 == 0 and c2.host_times()["repeats"] >= 1
            assert torch.equal(sim, sim2)
        finally:
            c2.close()
    finally:
        c.close()


@pytest.mark.parametrize("n,nr,ng,ebwt_on,mode", [
    (40_000_000, 100_000, 500, True, 0),      # configs[1]'s shape: 0.03 records per symbol, 50 MB table
    (40_000_000, 100_000, 500, True, 1),      # clustered: several times that
    (30_000_000, 400_000, 2000, False, 0),    # 800 MB table, EBWT=0
])
def test_density_probe_lands_the_first_pass(monkeypatch, n, nr, ng, ebwt_on, mode):
    """A fresh context's first pass (LiME_paired.sh:62-68 runs every collection once): the sampled probe's density is within a few % of
    what the pass then counts, the pass is not repeated, and its table is the oracle's.  (LIME_PROBE_MIN: by default only first passes of
    2^28 symbols and more are probed -- too long for the oracle; the shorter ones go binned with a pool for 0.45 records per symbol.)"""
    import torch
    import lime_amd
    monkeypatch.setenv("LIME_PROBE_MIN", str(1 << 24))
    lcp, da, eb = O.synth(5150 + mode, 0, n, nr, ng, 16, mode)
    e = eb if ebwt_on else None
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    exp = O.score(da, e, cl, nr, ng, threads=8)
    dev = torch.device("cuda", 0)
    tl = torch.from_numpy(lcp.view(np.int32)).to(dev); td = torch.from_numpy(da.view(np.int32)).to(dev)
    te = torch.from_numpy(eb).to(dev) if ebwt_on else None
    sim = torch.full((lime_amd.sim_bytes(nr, ng),), 0x33, dtype=torch.uint8, device=dev)
    c = lime_amd.Context()
    try:
        c.fused_dev(tl, td, te, n, n, True, nr, ng, 16, sim, True)
        probed = c.host_times()
        assert probed["probes"] == 1 and probed["records_per_symbol"] is not None
        s, rc = c.stats()
        assert rc == 0 and (s.n_clusters, s.max_len) == (nc, ml)
        true_density = s.n_updates / n
        assert abs(probed["records_per_symbol"] - true_density) <= 0.05 * true_density + 1e-4, (probed, true_density)
        ht = c.host_times()
        assert ht["repeats"] == 0 and ht["cas_fallbacks"] == 0 and not (s.flags & 128)
        assert np.array_equal(sim[:nr * ng].cpu().numpy().reshape(nr, ng), exp)
        # a second pass on the context: no second probe
        c.fused_dev(tl, td, te, n, n, True, nr, ng, 16, sim, True)
        s, rc = c.stats()
        assert rc == 0 and c.host_times()["probes"] == 1
        assert np.array_equal(sim[:nr * ng].cpu().numpy().reshape(nr, ng), exp)
    finally:
        c.close()


def test_probe_then_shards_and_streams_are_unchanged(monkeypatch):
    """the probe runs in front of whole passes only: shards of a stream (keep_stats) and passes that add to a table never see it"""
    import lime_amd
    monkeypatch.setenv("LIME_PROBE_MIN", str(1 << 24))
    c = lime_amd.Context()
    try:
        n, nr, ng = 18_000_000, 50_000, 400
        lcp, da, eb = O.synth(99, 0, n, nr, ng, 16, 1)
        cl, nc, ml = O.detect(lcp, da, nr, 16)
        exp = O.score(da, eb, cl, nr, ng, threads=8)
        sim, gnc, gml = c.fused_stream(lcp, da, eb, nr, ng, 16, chunk=5_000_000)
        assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp)
        assert c.host_times()["probes"] == 0
        sim, gnc, gml = c.fused(lcp, da, eb, nr, ng, 16)
        assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp)
    finally:
        c.close()


def test_fallback_to_compare_and_swap_is_loud(monkeypatch):
    """No device memory for the update records: the pass runs by compare-and-swap, gives the same table, and SAYS so -- LIME_FLAG_CAS_FALLBACK in
    the statistics, the reason in lime_last_error(), a count in lime_get_host_times.  (Provoked with a pool density no device can hold.)"""
    import lime_amd
    from lime_amd import _lib
    c = _ctx(monkeypatch, LIME_UPDATE_PATH="bin", LIME_POOL_DENSITY="30000")        # 17e6 symbols x 3e4 records x 4 bytes x 2 buffers > 288 GB
    try:
        n, nr, ng = 17_000_000, 50_000, 400
        lcp, da, eb = O.synth(123, 0, n, nr, ng, 16, 0)
        cl, nc, ml = O.detect(lcp, da, nr, 16)
        exp = O.score(da, eb, cl, nr, ng, threads=8)
        sim, gnc, gml = c.fused(lcp, da, eb, nr, ng, 16)
        assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp)
        s, rc = c.stats()
        assert rc == 0 and (s.flags & 128) and s.wave_records_max == 0
        assert c.host_times()["cas_fallbacks"] >= 1
        assert b"falls back to compare-and-swap" in _lib.load().lime_last_error()
    finally:
        c.close()


@pytest.mark.parametrize("shape", ["n1e10_clustered", "c5_clustered"])
def test_full_size_clustered_passes_stay_binned(shape):
    """N = 10^10 on one GPU at several times the iid generators' update density (the clustered generator; rounds 1-4 sent these passes to the
    compare-and-swap path: more records than 32-bit positions hold).  Size-independent properties: the pass takes the binned path with no
    fallback and no repeat after the probe; its table equals the table accumulated over eight position-range shards (each small enough for
    32-bit positions, each on the binned path too, added modulo 256 on the device); counters agree; the row scan is a plain reduction."""
    import torch
    import lime_amd
    from lime_amd.dist import shard_ranges, combine_edges
    n, nr, ng, ebwt_on = {"n1e10_clustered": (10_000_000_000, 1_000_000, 1000, False), "c5_clustered": (10_000_000_000, 3_000_000, 3423, True)}[shape]
    alpha, dev = 16, torch.device("cuda:0")
    lime_amd.trim_cache()
    c = lime_amd.Context()
    try:
        lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
        eb = torch.empty(n, dtype=torch.uint8, device=dev) if ebwt_on else None
        c.synth_dev(42, 0, n, nr, ng, alpha, 1, lcp, da, eb)
        tb = lime_amd.sim_bytes(nr, ng)
        A = torch.empty(tb, dtype=torch.uint8, device=dev)
        c.fused_dev(lcp, da, eb, n, n, True, nr, ng, alpha, A, True)
        sA, rc = c.stats(); assert rc == 0
        ht = c.host_times()
        assert sA.wave_records_max > 0 and not (sA.flags & 128), "the pass left the binned path"
        assert ht["probes"] == 1 and ht["repeats"] == 0 and ht["cas_fallbacks"] == 0, ht
        assert sA.n_updates > n // 10, int(sA.n_updates)             # clustered: well above the iid generators' 0.03 .. 0.12 per symbol
        B = torch.empty(tb, dtype=torch.uint8, device=dev)
        S = torch.zeros(tb, dtype=torch.uint8, device=dev)
        tot_c, tot_m, tot_u, edges = 0, 0, 0, []
        for lo, hi, hh in shard_ranges(n, 8):
            c.fused_dev(lcp[lo:], da[lo:], None if eb is None else eb[lo:], hi - lo, hh - lo, hh == n, nr, ng, alpha, B, True)
            s, rc = c.stats(); assert rc == 0 and s.wave_records_max > 0
            tot_c += s.n_clusters; tot_m = max(tot_m, s.max_len); tot_u += s.n_updates; edges.append(s.edge)
            S += B                                                   # uint8: modulo 256, like the ranks' reduce-scatter
        combine_edges(edges)
        assert (tot_c, tot_m, tot_u) == (sA.n_clusters, sA.max_len, sA.n_updates)
        assert torch.equal(A, S)
        del B, S
        mx = torch.empty(nr, dtype=torch.uint8, device=dev); nz = torch.empty(nr, dtype=torch.int32, device=dev)
        c.choose_dev(A, nr, ng, mx, nz)
        t2 = A[:nr * ng].view(nr, ng)
        assert torch.equal(mx, t2.amax(dim=1))
        assert int(nz.sum()) == int(torch.count_nonzero(t2))
    finally:
        c.close()


@pytest.mark.parametrize("wide", ["0", "1"])
@pytest.mark.parametrize("n", [12_000_000, 14_000_000, 17_700_000])
def test_bins_of_257_to_511_tiles(monkeypatch, n, wide):
    """One bin of 32 regions holding 336 / 392 / 495 second-level tiles: a wave of k_apply_tiles then walks 33 .. 63 runs per region, in steps of
    four, and the last step is short.  k_apply_tiles<true> took the index entries of a step's runs with __shfl inside a condition: a lane whose own run
    of the short last step does not exist was switched off, and whoever read from it got 0 -- groups 64 .. 79 of those runs were dropped without a
    trace (rounds 4: the N = 10^10 series has 308 tiles per bin; no test had a bin between 256 and 512 tiles).  Found by
    test_full_size_clustered_passes_stay_binned in round 5; this is the small case, against the oracle."""
    c = _ctx(monkeypatch, LIME_UPDATE_PATH="bin", LIME_BIN_LEVELS="1,1", LIME_APPLY_WIDE=wide)
    try:
        nr, ng = 2048, 1024
        lcp, da, _ = O.synth(77, 0, n, nr, ng, 16, 1)
        cl, nc, ml = O.detect(lcp, da, nr, 16)
        exp = O.score(da, None, cl, nr, ng, threads=8)
        sim, gnc, gml = c.fused(lcp, da, None, nr, ng, 16)
        s, rc = c.stats()
        assert rc == 0 and 256 * 8192 < s.n_updates < 512 * 8192, int(s.n_updates)
        assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp), int((sim != exp).sum())
    finally:
        c.close()


@pytest.mark.parametrize("mode,ebwt_on", [(0, True), (1, True), (1, False)])
def test_first_pass_without_probe_is_binned_and_right(mode, ebwt_on):
    """first passes below 2^28 symbols run without the probe: binned, a pool for 0.45 records per symbol -- no repeat at the generators' 0.03 .. 0.23 --,
    and later passes choose their path from the density the first one counted"""
    import lime_amd
    n, nr, ng = 20_000_000, 100_000, 500
    lcp, da, eb = O.synth(61 + mode, 0, n, nr, ng, 16, mode)
    e = eb if ebwt_on else None
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    exp = O.score(da, e, cl, nr, ng, threads=8)
    c = lime_amd.Context()
    try:
        for k in range(2):
            sim, gnc, gml = c.fused(lcp, da, e, nr, ng, 16)
            s, rc = c.stats()
            assert rc == 0 and (gnc, gml) == (nc, ml) and np.array_equal(sim, exp)
            if k == 0:
                assert s.wave_records_max > 0, "an unmeasured first pass takes the binned path"
        ht = c.host_times()
        assert ht["probes"] == 0 and ht["repeats"] == 0 and ht["cas_fallbacks"] == 0, ht
    finally:
        c.close()


@pytest.mark.parametrize("nr,ng", [(1_000_000, 1000), (1_000_000, 4000)])
def test_more_than_2_32_records_in_one_pass(nr, ng):
    """1.9 * 10^10 symbols of the clustered generator on one GPU (152 GB of arrays): 4.4 * 10^9 update records -- more than 32-bit positions hold -- in ONE
    pass on the binned path: k_part_lines<true> (1 GB table, 477 bins) and k_part<.., true> (4 GB table, 954 bins) with positions whose high word is 1
    for real.  The table == the table of the compare-and-swap path of the same library (the oracle cannot go there), counters equal, no repeat,
    no fallback.  Reference: the updates of ClusterBWT_DA.cpp:178-184, 243-248 at the density of real text and the size of configs[4]."""
    import os
    import torch
    import lime_amd
    torch.cuda.empty_cache()
    lime_amd.trim_cache()
    free, total = torch.cuda.mem_get_info()
    n = 19_000_000_000
    if free < (8 * n + 2 * nr * ng + 70 * (1 << 30)):
        pytest.skip(f"needs about {(8 * n + 2 * nr * ng) / 1e9 + 70:.0f} GB of free HBM, {free / 1e9:.0f} GB are free")
    dev = torch.device("cuda", 0)
    c = lime_amd.Context()
    old = os.environ.get("LIME_UPDATE_PATH")
    os.environ["LIME_UPDATE_PATH"] = "cas"
    c0 = lime_amd.Context()
    if old is None:
        del os.environ["LIME_UPDATE_PATH"]
    else:
        os.environ["LIME_UPDATE_PATH"] = old
    try:
        lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
        c.synth_dev(42, 0, n, nr, ng, 16, 1, lcp, da, None)
        tb = lime_amd.sim_bytes(nr, ng)
        A = torch.empty(tb, dtype=torch.uint8, device=dev)
        c.fused_dev(lcp, da, None, n, n, True, nr, ng, 16, A, True)
        sA, rc = c.stats(); assert rc == 0
        ht = c.host_times()
        assert sA.n_updates > (1 << 32), int(sA.n_updates)
        assert sA.wave_records_max > 0 and not (sA.flags & 128) and ht["repeats"] == 0 and ht["cas_fallbacks"] == 0, (int(sA.flags), ht)
        T = torch.empty(tb, dtype=torch.uint8, device=dev)
        c0.fused_dev(lcp, da, None, n, n, True, nr, ng, 16, T, True)
        sT, rc = c0.stats(); assert rc == 0 and sT.wave_records_max == 0
        assert (sA.n_clusters, sA.max_len, sA.n_updates) == (sT.n_clusters, sT.max_len, sT.n_updates)
        assert torch.equal(A, T), int((A != T).sum())
    finally:
        c.close(); c0.close()
        torch.cuda.empty_cache()


@pytest.mark.parametrize("group", ["", "1", "2", "3", "4", "5", "6"])
@pytest.mark.parametrize("nr,ng,n", [(8192, 1024, 6_000_000), (16384, 2048, 5_000_000), (3000, 1400, 9_000_000)])
def test_short_runs_of_a_region_share_a_wave(monkeypatch, nr, ng, n, group):
    """k_apply_tiles on bins of many regions (one bin of 128 / 512 / 65 regions: LIME_BIN_LEVELS=1,1): a region's run in a tile row is 64 / 16 / 126
    records on average, and 2^lg lanes share a run, two groups of four records a lane (round 6; before, a wave per run).  Every group size (option
    apply_group: 2 .. 64 lanes; "" = by the regions per bin) against the oracle: runs shorter and longer than a pass takes, empty runs, the last
    short step of a wave's runs, the bins' last, partly filled tile row."""
    c = _ctx(monkeypatch, LIME_UPDATE_PATH="bin", LIME_BIN_LEVELS="1,1", LIME_APPLY_GROUP=group)
    try:
        lcp, da, _ = O.synth(1234 + ng, 0, n, nr, ng, 16, 1)
        cl, nc, ml = O.detect(lcp, da, nr, 16)
        exp = O.score(da, None, cl, nr, ng, threads=8)
        sim, gnc, gml = c.fused(lcp, da, None, nr, ng, 16)
        s, rc = c.stats()
        assert rc == 0 and s.n_updates > 20 * 8192, int(s.n_updates)
        assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp), int((sim != exp).sum())
    finally:
        c.close()
        _ctx(monkeypatch, LIME_APPLY_GROUP="").close()                # (the option is process-wide: back to the library's choice)
