"""clusterChoose without the table (round 5, lime_fused_choose_dev): the pass stops at the binned update records, k_apply_tiles
builds every 64 KB region of the table in LDS and hands over row maxima / non-zero counts, then the passing rows' (idRef, sim) lists -- the table is
never written.  Against the oracle's table + numpy, and (full size) against the table path of the same library.  Reference: ClusterBWT_DA.cpp:385-423."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


def _expected(sim, norm, beta):
    mx = sim.max(axis=1) if sim.shape[1] else np.zeros(sim.shape[0], np.uint8)
    ok = (mx.astype(np.float32) / np.float32(norm)) > np.float32(beta)
    off = np.zeros(sim.shape[0] + 1, np.uint64)
    rows = []
    for r in range(sim.shape[0]):
        if ok[r]:
            nzc = np.nonzero(sim[r])[0]
            rows.append(np.stack([nzc.astype(np.uint32), sim[r][nzc].astype(np.uint32)], axis=1))
            off[r + 1] = off[r] + len(nzc)
        else:
            off[r + 1] = off[r]
    pairs = np.concatenate(rows) if rows else np.zeros((0, 2), np.uint32)
    return mx, off, pairs


def _run(monkeypatch, lcp, da, eb, nr, ng, norm, beta, **env):
    import torch
    import lime_amd
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    c = lime_amd.Context()
    try:
        dev = torch.device("cuda", 0)
        tl = torch.from_numpy(lcp.view(np.int32)).to(dev); td = torch.from_numpy(da.view(np.int32)).to(dev)
        te = None if eb is None else torch.from_numpy(eb).to(dev)
        mx, off, pairs, s = c.fused_choose_dev(tl, td, te, len(lcp), nr, ng, 16, norm, beta)
        s.table_free = c.host_times()["choose_without_table"]
        return mx, off, pairs, s
    finally:
        c.close()


@pytest.mark.parametrize("wide", ["0", "1"])
@pytest.mark.parametrize("levels,nr,ng,mode", [
    ("1,2", 3000, 300, 1),            # 14 regions in one bin; rows of 300 bytes: 219 segments per region
    ("2,3", 5000, 1200, 1),           # 3 bins of 32 regions
    ("4,7", 40000, 700, 0),           # 428 regions
    ("1,1", 100000, 3, 1),            # rows of 3 bytes: 21 846 segments per region
    ("1,2", 7, 131072, 1),            # rows of two regions each: cells of a row in earlier regions come first in its list
    ("1,2", 5, 200000, 0),            # rows of 3.05 regions, borders anywhere
    ("1,2", 1000, 65536, 1),          # rows = regions
    ("1,1", 129, 1027, 1),            # odd width, the table ends inside a region
    (None, 150000, 3000, 0),          # the default layout of a 450 MB table: 430 bins of 16 regions
])
def test_choose_without_the_table_vs_oracle(monkeypatch, levels, nr, ng, mode, wide):
    n = 1_500_000
    lcp, da, eb = O.synth(8100 + nr, 0, n, nr, ng, 16, mode)
    lcp[n // 3:n // 3 + 700] = 40                                  # a long cluster: its updates come as records of their own, bucketed by region
    lcp[n // 2:n // 2 + 90] = 33
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    env = {"LIME_UPDATE_PATH": "bin", "LIME_CHOOSE_FREE": 1, "LIME_APPLY_WIDE": wide}
    if levels:
        env["LIME_BIN_LEVELS"] = levels
    for e, beta in ((eb, 0.02), (None, 0.012), (eb, 0.0)):
        sim = O.score(da, e, cl, nr, ng, threads=8)
        emx, eoff, epairs = _expected(sim, 85, beta)
        mx, off, pairs, s = _run(monkeypatch, lcp, da, e, nr, ng, 85, beta, **env)
        assert (s.n_clusters, s.max_len) == (nc, ml) and s.wave_records_max > 0 and s.table_free == 1
        assert np.array_equal(mx, emx), int((mx != emx).sum())
        assert np.array_equal(off, eoff)
        assert np.array_equal(pairs, epairs), (len(pairs), len(epairs))


@pytest.mark.parametrize("ebwt_mode", [1, 0])
def test_choose_without_the_table_writes_reference_files(monkeypatch, golden, ebwt_mode, tmp_path):
    """the goldens through lime_fused_choose_dev (second level forced where the table has more than one region; the one-region tables take the
    table path of the same call): .res.txt / .res.bin / .res.pos bytes of the reference"""
    import lime_amd
    g = golden
    nr, ng = g["n_reads"], g["n_refs"]
    norm, beta = g["read_len"] + 1 - g["alpha"], float(g["beta"])
    eb = g["ebwt"] if ebwt_mode else None
    if len(g["lcp"]) == 0:
        pytest.skip("empty collection")
    mx, off, pairs, s = _run(monkeypatch, g["lcp"].astype(np.uint32), g["da"].astype(np.uint32), eb, nr, ng, norm, beta,
                             LIME_UPDATE_PATH="bin", LIME_CHOOSE_FREE=1, LIME_BIN_LEVELS="1,2")
    assert s.n_clusters == len(g["clrs"])
    lib = lime_amd._lib.load()
    pr = np.ascontiguousarray(pairs); mxa, offa = np.ascontiguousarray(mx), np.ascontiguousarray(off)
    t, b, q = (str(tmp_path / k).encode() for k in ("r.txt", "r.bin", "r.pos"))
    assert lib.lime_write_res_txt_pairs(t, mxa.ctypes.data, offa.ctypes.data, pr.ctypes.data, nr, norm, C.c_float(beta)) == 0
    assert lib.lime_write_res_bin_pairs(b, q, mxa.ctypes.data, offa.ctypes.data, pr.ctypes.data, nr, norm, C.c_float(beta)) == 0
    assert open(t, "rb").read() == g[f"txt_e{ebwt_mode}"].tobytes()
    assert open(b, "rb").read() == g[f"bin_e{ebwt_mode}"].tobytes()
    assert open(q, "rb").read() == g[f"pos_e{ebwt_mode}"].tobytes()


def test_choose_without_the_table_wraps_modulo_256(monkeypatch):
    """a table of three regions in which a handful of cells takes every update: they pass 255 many times -- the regions are rebuilt with the exact
    adds before they are looked at"""
    rng = np.random.default_rng(17)
    n, nr, ng = 900_000, 70_000, 2
    lcp = np.where(rng.random(n) < 0.6, 20, 3).astype(np.uint32); lcp[0] = 0
    reads = np.array([0, 1, 32768, 69_999], np.uint32)             # rows in the first, second and last region
    da = np.where(rng.random(n) < 0.4, reads[rng.integers(0, 4, n)], nr + rng.integers(0, 2, n)).astype(np.uint32)
    eb = rng.choice(np.frombuffer(b"ACGTN", np.uint8), n).astype(np.uint8)
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    for e in (eb, None):
        sim = O.score(da, e, cl, nr, ng, threads=4)
        assert int(sim.astype(np.int64).sum()) > 0
        for beta in (0.0, 0.5, 2.0):
            emx, eoff, epairs = _expected(sim, 85, beta)
            mx, off, pairs, s = _run(monkeypatch, lcp, da, e, nr, ng, 85, beta, LIME_UPDATE_PATH="bin", LIME_CHOOSE_FREE=1, LIME_BIN_LEVELS="1,1")
            assert s.n_updates > 256 * 8 and s.wave_records_max > 0 and s.table_free == 1
            assert np.array_equal(mx, emx) and np.array_equal(off, eoff) and np.array_equal(pairs, epairs)


@pytest.mark.parametrize("shape", ["C3", "C2x"])
def test_choose_without_the_table_at_full_size(monkeypatch, shape):
    """configs[2] (10^9 symbols, 10^6 x 5000, EBWT=0) and a configs[1]-like EBWT=1 pass over a 1.5 GB table: the lists of the call without the
    table == the lists of table + k_choose + k_gather_pairs (option choose_free 0), for a beta that lets a part of the rows pass"""
    import torch
    import lime_amd
    n, nr, ng, ebwt_on, mode = {"C3": (1_000_000_000, 1_000_000, 5000, False, 0), "C2x": (300_000_000, 500_000, 3000, True, 1)}[shape]
    dev = torch.device("cuda", 0)
    c = lime_amd.Context()
    lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
    eb = torch.empty(n, dtype=torch.uint8, device=dev) if ebwt_on else None
    c.synth_dev(42, 0, n, nr, ng, 16, mode, lcp, da, eb)
    res = []
    try:
        for free in ("1", "0"):
            c.set_option("choose_free", free)
            mx, off, pairs, s = c.fused_choose_dev(lcp, da, eb, n, nr, ng, 16, 85, 0.03)      # max >= 3 passes
            res.append((mx, off, pairs, s.n_updates, s.n_clusters))
        assert c.host_times()["choose_without_table"] == 1
        # the caller's result arrays (`out=`), dirty on entry: the same rows, and the pair list is the library's buffer itself (no copy)
        outs = (np.full(nr + 1, 0xAB, dtype=np.uint8), np.full(nr + 2, 0xCDCDCDCD, dtype=np.uint64))
        mx2, off2, pairs2, _ = c.fused_choose_dev(lcp, da, eb, n, nr, ng, 16, 85, 0.03, out=outs)
        assert mx2.base is outs[0] and off2.base is outs[1] and pairs2.base is not None
        assert np.array_equal(mx2, res[1][0]) and np.array_equal(off2, res[1][1]) and np.array_equal(pairs2, res[1][2])
    finally:
        c.close()
    assert res[0][3] == res[1][3] and res[0][4] == res[1][4]
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    assert 0 < len(res[0][2]) and int((np.diff(res[0][1].astype(np.int64)) > 0).sum()) < nr      # some rows pass, not all


@pytest.mark.parametrize("group", ["", "2", "3", "4", "6"])
def test_choose_without_the_table_on_hot_rows(monkeypatch, group):
    """All reads in 50 rows of a table whose rows are regions (1000 x 65536, two bins of 512 regions): a region of a hot row has thousands of cells and
    long runs in every tile row, so k_apply_tiles<., 1> queues hundreds of first adds per wave and its run loop takes several passes per run -- with 2^lg
    lanes a run (round 6) lanes without further groups leave that loop on their own, and while the queue's count was a per-lane copy they came back with
    a stale one and wrote over queued cells (60 % of a hot row's cells were not counted).  Row maxima, counts and lists against the oracle's table."""
    nr, ng, n = 1000, 65536, 1_500_000
    lcp, da, eb = O.synth(8100 + nr, 0, n, nr, ng, 16, 1)
    rd = da < nr
    da[rd] = da[rd] % 50
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    sim = O.score(da, eb, cl, nr, ng, threads=8)
    emx, eoff, epairs = _expected(sim, 85, 0.0)
    try:
        mx, off, pairs, s = _run(monkeypatch, lcp, da, eb, nr, ng, 85, 0.0, LIME_UPDATE_PATH="bin", LIME_CHOOSE_FREE=1, LIME_BIN_LEVELS="1,2", LIME_APPLY_GROUP=group)
        assert (s.n_clusters, s.max_len) == (nc, ml) and s.table_free == 1
        assert np.array_equal(mx, emx) and np.array_equal(off, eoff) and np.array_equal(pairs, epairs)
        assert int(np.diff(eoff.astype(np.int64)).max()) > 3000              # (the hot rows are what this is about)
    finally:
        _run(monkeypatch, lcp[:4096], da[:4096], eb[:4096], nr, ng, 85, 0.5, LIME_APPLY_GROUP="")      # (the option is process-wide: back to the library's choice)
