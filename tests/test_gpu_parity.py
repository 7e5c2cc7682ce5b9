"""Parity of the HIP path (through the C ABI) with the oracle and the reference's golden
vectors.  Integer/byte work: every comparison is bit-exact."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle_py as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "lime_amd", "bin")
GOLDEN_FIRST = "edges"                    # (the golden case whose turn also runs the synthetic part of test_dense_windows_...)


@pytest.fixture(scope="module")
def ctx():
    import lime_amd
    c = lime_amd.Context()
    yield c
    c.close()


# ---- golden vectors of the reference -----------------------------------------------------
def test_detect_golden(ctx, golden):
    cl, nc, ml = ctx.detect(golden["lcp"], golden["da"], golden["n_reads"], golden["alpha"])
    assert nc == len(golden["clrs"])
    assert np.array_equal(cl, golden["clrs"])
    assert O.out_bytes(golden["n_reads"], golden["n_refs"], golden["alpha"], ml, nc) == golden["out"].tobytes()


def test_detect_golden_in_chunks(ctx, golden, monkeypatch):
    """lime_detect walks the arrays in position-range chunks (here 4096 positions): same records, same order"""
    monkeypatch.setenv("LIME_DETECT_CHUNK", "4096")
    monkeypatch.setenv("LIME_FORCE_STAGING", "1")           # through the pinned staging ring although the arrays are small
    cl, nc, ml = ctx.detect(golden["lcp"], golden["da"], golden["n_reads"], golden["alpha"])
    assert nc == len(golden["clrs"]) and np.array_equal(cl, golden["clrs"])
    assert ml == (int(golden["clrs"][:, 1].max()) if nc else 0)


@pytest.mark.parametrize("ebwt_mode", [1, 0])
def test_score_golden(ctx, golden, ebwt_mode):
    eb = golden["ebwt"] if ebwt_mode else None
    sim = ctx.score(golden["da"], eb, golden["clrs"], golden["n_reads"], golden["n_refs"])
    assert np.array_equal(sim, golden[f"sim_e{ebwt_mode}"])


@pytest.mark.parametrize("ebwt_mode", [1, 0])
def test_fused_golden(ctx, golden, ebwt_mode):
    eb = golden["ebwt"] if ebwt_mode else None
    sim, nc, ml = ctx.fused(golden["lcp"], golden["da"], eb, golden["n_reads"], golden["n_refs"], golden["alpha"])
    assert nc == len(golden["clrs"])
    assert ml == (int(golden["clrs"][:, 1].max()) if nc else 0)
    assert np.array_equal(sim, golden[f"sim_e{ebwt_mode}"])


@pytest.mark.parametrize("ebwt_mode", [1, 0])
def test_fused_stream_golden(ctx, golden, ebwt_mode):
    """the chunked host-streaming entry point (one 4096-position chunk at a time) = lime_fused"""
    eb = golden["ebwt"] if ebwt_mode else None
    sim, nc, ml = ctx.fused_stream(golden["lcp"], golden["da"], eb, golden["n_reads"], golden["n_refs"],
                                   golden["alpha"], chunk=4096)
    assert nc == len(golden["clrs"])
    assert ml == (int(golden["clrs"][:, 1].max()) if nc else 0)
    assert np.array_equal(sim, golden[f"sim_e{ebwt_mode}"])


@pytest.mark.parametrize("n,chunk", [(1, 4096), (4096, 4096), (4097, 4096), (300001, 8192), (300001, 65536),
                                     (1000003, 0), (1000003, 262144)])
def test_fused_stream_synth_vs_oracle(ctx, n, chunk, monkeypatch):
    if n == 300001:
        monkeypatch.setenv("LIME_FORCE_STAGING", "1")       # dozens of chunks through the three pinned slots
    nr, ng, alpha = 500, 40, 16
    lcp, da, eb = O.synth(77, 0, n, nr, ng, alpha, 1)
    clrs, _, ml = O.detect(lcp, da, nr, alpha)
    exp = O.score(da, eb, clrs, nr, ng, threads=4)
    sim, nc, gml = ctx.fused_stream(lcp, da, eb, nr, ng, alpha, chunk=chunk)
    assert nc == len(clrs) and gml == ml
    assert np.array_equal(sim, exp)


def test_score_in_chunks(ctx, golden, monkeypatch):
    """lime_score takes the arrays through HBM in position-range chunks (here 4096 positions), clusters in
    any order"""
    monkeypatch.setenv("LIME_SCORE_CHUNK", "4096")
    monkeypatch.setenv("LIME_FORCE_STAGING", "1")
    rng = np.random.default_rng(5)
    cl = golden["clrs"][rng.permutation(len(golden["clrs"]))]
    for mode in (1, 0):
        sim = ctx.score(golden["da"], golden["ebwt"] if mode else None, cl, golden["n_reads"], golden["n_refs"])
        assert np.array_equal(sim, golden[f"sim_e{mode}"])


def test_score_accepts_any_cluster_order(ctx, golden):
    rng = np.random.default_rng(3)
    cl = golden["clrs"][rng.permutation(len(golden["clrs"]))]
    sim = ctx.score(golden["da"], golden["ebwt"], cl, golden["n_reads"], golden["n_refs"])
    assert np.array_equal(sim, golden["sim_e1"])


def test_choose_golden(ctx, golden):
    sim = golden["sim_e1"]
    mx, nz = ctx.choose(sim)
    emx, enz = O.choose(sim)
    assert np.array_equal(mx, emx) and np.array_equal(nz, enz)


@pytest.mark.parametrize("ebwt_mode", [1, 0])
def test_score_choose_writes_reference_files(ctx, golden, ebwt_mode, tmp_path):
    """clusterAnalyze + clusterChoose with the table left on the device: the compact (row_max,
    row_off, pairs) form written by the pair writers = the reference's .res.txt / .res.bin / .res.pos."""
    import ctypes as C
    eb = golden["ebwt"] if ebwt_mode else None
    nr, ng = golden["n_reads"], golden["n_refs"]
    norm, beta = golden["read_len"] + 1 - golden["alpha"], float(golden["beta"])
    mx, off, pairs, sim = ctx.score_choose(golden["da"], eb, golden["clrs"], nr, ng, norm, beta, want_sim=True)
    assert np.array_equal(sim, golden[f"sim_e{ebwt_mode}"])
    emx, enz = O.choose(sim)
    assert np.array_equal(mx, emx)
    # the lists: non-zero cells of passing rows, ascending idRef
    for r in range(nr):
        row = pairs[int(off[r]):int(off[r + 1])]
        if np.float32(mx[r]) / np.float32(norm) > np.float32(beta):
            nzc = np.nonzero(sim[r])[0]
            assert np.array_equal(row[:, 0], nzc) and np.array_equal(row[:, 1], sim[r][nzc])
        else:
            assert len(row) == 0
    lib = ctx.lib
    pr = np.ascontiguousarray(pairs)
    mxa, offa = np.ascontiguousarray(mx), np.ascontiguousarray(off)
    t, b, q = (str(tmp_path / n).encode() for n in ("r.txt", "r.bin", "r.pos"))
    assert lib.lime_write_res_txt_pairs(t, mxa.ctypes.data, offa.ctypes.data, pr.ctypes.data, nr, norm, C.c_float(beta)) == 0
    assert lib.lime_write_res_bin_pairs(b, q, mxa.ctypes.data, offa.ctypes.data, pr.ctypes.data, nr, norm, C.c_float(beta)) == 0
    assert open(t, "rb").read() == golden[f"txt_e{ebwt_mode}"].tobytes()
    assert open(b, "rb").read() == golden[f"bin_e{ebwt_mode}"].tobytes()
    assert open(q, "rb").read() == golden[f"pos_e{ebwt_mode}"].tobytes()


@pytest.mark.parametrize("n_reads,n_refs", [(1, 1), (7, 3), (300, 1), (129, 257), (1000, 930), (5, 5000)])
def test_choose_pairs_dev_random_tables(ctx, n_reads, n_refs):
    """compaction of a device-resident table vs numpy: ragged rows, widths that are not
    multiples of 4, empty and full rows."""
    import torch
    rng = np.random.default_rng(n_reads * 7919 + n_refs)
    sim = (rng.integers(0, 256, (n_reads, n_refs)) * (rng.random((n_reads, n_refs)) < 0.1)).astype(np.uint8)
    if n_reads > 2:
        sim[1] = 0
        sim[2] = rng.integers(1, 256, n_refs)
    import lime_amd
    buf = torch.zeros(lime_amd.sim_bytes(n_reads, n_refs), dtype=torch.uint8, device="cuda:0")
    buf[:n_reads * n_refs] = torch.from_numpy(sim.reshape(-1)).to("cuda:0")
    norm, beta = 85, 0.25
    mx, off, pairs = ctx.choose_pairs_dev(buf, n_reads, n_refs, norm, beta)
    assert np.array_equal(mx, sim.max(axis=1))
    total = 0
    for r in range(n_reads):
        row = pairs[int(off[r]):int(off[r + 1])]
        if np.float32(mx[r]) / np.float32(norm) > np.float32(beta):
            nzc = np.nonzero(sim[r])[0]
            assert np.array_equal(row[:, 0], nzc) and np.array_equal(row[:, 1], sim[r][nzc])
            total += len(nzc)
        else:
            assert len(row) == 0
    assert total == len(pairs) == int(off[n_reads])


# ---- the drop-in executables write the reference's bytes -----------------------------------
@pytest.mark.parametrize("ebwt_mode,binary", [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_cli_dropin_files(golden, ebwt_mode, binary, tmp_path):
    g = golden
    base = str(tmp_path / "X.fasta")
    g["lcp"].astype("<u4").tofile(base + ".lcp")
    g["da"].astype("<u4").tofile(base + ".da")
    g["ebwt"].tofile(base + ".ebwt")
    env = dict(os.environ, LIME_EBWT=str(ebwt_mode), LIME_BIN=str(binary))
    r = subprocess.run([f"{BIN}/ClusterLCP", base, str(g["n_reads"]), str(g["n_refs"]), str(g["alpha"]), "4"],
                       capture_output=True, timeout=300, env=env, cwd=tmp_path)
    assert r.returncode == 0, r.stderr.decode()
    assert open(f"{base}.{g['alpha']}.clrs", "rb").read() == g["clrs"].tobytes()
    assert open(str(tmp_path / "X.out"), "rb").read() == g["out"].tobytes()
    r = subprocess.run([f"{BIN}/ClusterBWT_DA", base, str(g["read_len"]), repr(g["beta"]), "4"],
                       capture_output=True, timeout=300, env=env, cwd=tmp_path)
    assert r.returncode == 0, r.stderr.decode()
    tag = f"e{ebwt_mode}"
    if binary:
        assert open(base + ".res.bin", "rb").read() == g[f"bin_{tag}"].tobytes()
        assert open(base + ".res.pos", "rb").read() == g[f"pos_{tag}"].tobytes()
    else:
        assert open(base + ".res.txt", "rb").read() == g[f"txt_{tag}"].tobytes()


@pytest.mark.parametrize("ebwt_mode,binary", [(1, 1), (0, 0)])
def test_cli_dropin_multi_gpu_entry(golden, ebwt_mode, binary, tmp_path):
    """ClusterBWT_DA under LIME_GPUS=1 goes through lime_score_choose_multi (cluster list cut by position, row blocks of
    a multiple of 16 rows, RCCL reduce-scatter -- forced here on the one device -- and per-block row scans): same bytes"""
    g = golden
    base = str(tmp_path / "X.fasta")
    g["lcp"].astype("<u4").tofile(base + ".lcp"); g["da"].astype("<u4").tofile(base + ".da"); g["ebwt"].tofile(base + ".ebwt")
    g["clrs"].astype("<u8").tofile(f"{base}.{g['alpha']}.clrs")
    open(str(tmp_path / "X.out"), "wb").write(g["out"].tobytes())
    env = dict(os.environ, LIME_EBWT=str(ebwt_mode), LIME_BIN=str(binary), LIME_GPUS="1", LIME_FORCE_RCCL="1")
    r = subprocess.run([f"{BIN}/ClusterBWT_DA", base, str(g["read_len"]), repr(g["beta"]), "4"],
                       capture_output=True, timeout=300, env=env, cwd=tmp_path)
    assert r.returncode == 0, r.stderr.decode()
    tag = f"e{ebwt_mode}"
    if binary:
        assert open(base + ".res.bin", "rb").read() == g[f"bin_{tag}"].tobytes()
        assert open(base + ".res.pos", "rb").read() == g[f"pos_{tag}"].tobytes()
    else:
        assert open(base + ".res.txt", "rb").read() == g[f"txt_{tag}"].tobytes()


def test_cli_usage_errors(tmp_path):
    r = subprocess.run([f"{BIN}/ClusterLCP", "x"], capture_output=True, timeout=60)
    assert r.returncode == 1 and b"Error usage" in r.stderr
    r = subprocess.run([f"{BIN}/ClusterBWT_DA", "x"], capture_output=True, timeout=60)
    assert r.returncode == 1 and b"Error usage" in r.stderr
    r = subprocess.run([f"{BIN}/ClusterLCP", str(tmp_path / "nofile.fasta"), "1", "1", "16", "1"],
                       capture_output=True, timeout=60)
    assert r.returncode != 0 and b"Error opening" in r.stderr


# ---- python mirror of the two programs ------------------------------------------------------
def test_python_program_mirror(ctx, golden, tmp_path):
    import lime_amd
    g = golden
    base = str(tmp_path / "Y.fasta")
    g["lcp"].astype("<u4").tofile(base + ".lcp")
    g["da"].astype("<u4").tofile(base + ".da")
    g["ebwt"].tofile(base + ".ebwt")
    lime_amd.cluster_lcp(base, g["n_reads"], g["n_refs"], g["alpha"], ctx=ctx)
    assert open(f"{base}.{g['alpha']}.clrs", "rb").read() == g["clrs"].tobytes()
    lime_amd.cluster_bwt_da(base, g["read_len"], g["beta"], ebwt=True, binary=False, ctx=ctx)
    assert open(base + ".res.txt", "rb").read() == g["txt_e1"].tobytes()
    lime_amd.cluster_bwt_da(base, g["read_len"], g["beta"], ebwt=False, binary=True, ctx=ctx)
    assert open(base + ".res.bin", "rb").read() == g["bin_e0"].tobytes()
    assert open(base + ".res.pos", "rb").read() == g["pos_e0"].tobytes()


# ---- seeded synthetic inputs against the oracle ---------------------------------------------
@pytest.mark.parametrize("n,nr,ng,mode", [
    (1, 3, 2, 0), (2, 3, 2, 0), (63, 3, 2, 0), (4095, 5, 4, 0), (4096, 5, 4, 1), (4097, 5, 4, 0),
    (8192, 40, 7, 1), (12289, 40, 7, 0), (300000, 1000, 50, 0), (300001, 200, 9, 1), (2000003, 5000, 120, 0),
])
def test_fused_vs_oracle_synth(ctx, n, nr, ng, mode):
    lcp, da, eb = O.synth(1000 + n, 0, n, nr, ng, 16, mode)
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    for e in (eb, None):
        exp = O.score(da, e, cl, nr, ng, threads=4)
        sim, gnc, gml = ctx.fused(lcp, da, e, nr, ng, 16)
        assert (gnc, gml) == (nc, ml)
        assert np.array_equal(sim, exp)
    gcl, gnc, gml = ctx.detect(lcp, da, nr, 16)
    assert np.array_equal(gcl, cl) and (gnc, gml) == (nc, ml)


def test_truncated_lcp_gives_the_same_clusters_and_table(ctx):
    """README.md:59-61 (eGap --trlcp k, k >= alpha): an lcp array capped at k is as good as the full one"""
    from tests.conftest import load_golden
    g = load_golden("text_example")
    for k in (g["alpha"], 40):
        lcp = np.minimum(g["lcp"], k).astype(np.uint32)
        cl, nc, ml = ctx.detect(lcp, g["da"], g["n_reads"], g["alpha"])
        assert np.array_equal(cl, g["clrs"])
        sim, gnc, gml = ctx.fused(lcp, g["da"], g["ebwt"], g["n_reads"], g["n_refs"], g["alpha"])
        assert gnc == len(g["clrs"]) and np.array_equal(sim, g["sim_e1"])


def test_empty_input(ctx):
    z32 = np.zeros(0, np.uint32)
    cl, nc, ml = ctx.detect(z32, z32, 3, 16)
    assert nc == 0 and ml == 0 and cl.shape == (0, 2)
    sim, nc, ml = ctx.fused(z32, z32, np.zeros(0, np.uint8), 3, 2, 16)
    assert nc == 0 and not sim.any()
    sim = ctx.score(z32, None, np.zeros((0, 2), np.uint64), 3, 2)
    assert not sim.any()


@pytest.mark.parametrize("alpha", [1, 16, 40, 63])
def test_alpha_sweep(ctx, alpha):
    lcp, da, eb = O.synth(77, 0, 200000, 300, 11, 16, 0)
    cl, nc, ml = O.detect(lcp, da, 300, alpha)
    exp = O.score(da, eb, cl, 300, 11, threads=4)
    sim, gnc, gml = ctx.fused(lcp, da, eb, 300, 11, alpha)
    assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp)


def test_long_clusters_and_tile_crossings(ctx):
    """runs of every scale: > 64 (big-cluster kernel), > 4096 (several tiles), reaching EOF;
    few documents so per-cluster counts pass 255 (genome saturation, read wrap)."""
    rng = np.random.default_rng(11)
    n = 200000
    lcp = np.where(rng.random(n) < 0.97, 20, 3).astype(np.uint32)
    lcp[0] = 0
    lcp[20000:45000] = 30          # 25000-long run
    lcp[45000] = 1
    lcp[100000:100000 + 65536] = 30  # exactly at the 65536 limit
    lcp[100000] = 0
    lcp[100000 + 65536] = 0
    da = np.where(rng.random(n) < 0.5, rng.integers(0, 3, n), 3 + rng.integers(0, 4, n)).astype(np.uint32)
    eb = rng.choice(np.frombuffer(b"ACGTNRY\x00", np.uint8), n).astype(np.uint8)
    cl, nc, ml = O.detect(lcp, da, 3, 16)
    assert ml == 65536
    for e in (eb, None):
        exp = O.score(da, e, cl, 3, 4, threads=4)
        sim, gnc, gml = ctx.fused(lcp, da, e, 3, 4, 16)
        assert (gnc, gml) == (nc, ml)
        assert np.array_equal(sim, exp)
        assert np.array_equal(ctx.score(da, e, cl, 3, 4), exp)


def test_many_docs_big_cluster(ctx):
    """a long cluster with thousands of distinct reads and genomes (hash table path)."""
    rng = np.random.default_rng(12)
    n = 30000
    lcp = np.full(n, 25, np.uint32); lcp[0] = 0; lcp[20000] = 2; lcp[20001:] = 3
    nr, ng = 3000, 700
    da = np.where(rng.random(n) < 0.6, rng.integers(0, nr, n), nr + rng.integers(0, ng, n)).astype(np.uint32)
    eb = rng.choice(np.frombuffer(b"ACGTN", np.uint8), n).astype(np.uint8)
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    assert nc == 1 and ml == 20000
    exp = O.score(da, eb, cl, nr, ng)
    sim, gnc, gml = ctx.fused(lcp, da, eb, nr, ng, 16)
    assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp)


def test_maxlen_error(ctx):
    import lime_amd
    n = 70000
    lcp = np.full(n, 20, np.uint32); lcp[0] = 0
    da = (np.arange(n) % 2).astype(np.uint32)
    cl, nc, ml = ctx.detect(lcp, da, 1, 16)           # detection has no limit (ClusterLCP)
    assert nc == 1 and ml == n
    with pytest.raises(lime_amd.LimeError) as e:       # scoring refuses > 65536 (ClusterBWT_DA.cpp:558-562)
        ctx.fused(lcp, da, None, 1, 1, 16)
    assert e.value.code == -4
    with pytest.raises(lime_amd.LimeError):
        ctx.score(da, None, cl, 1, 1)


def test_detect_in_chunks_with_a_run_longer_than_the_halo(ctx, monkeypatch):
    """ClusterLCP has no length limit: a run longer than the chunk halo makes the chunked walk start over as
    one chunk instead of failing"""
    monkeypatch.setenv("LIME_DETECT_CHUNK", "4096")
    n = 200000
    lcp, da, _ = O.synth(5, 0, n, 50, 7, 16, 0)
    lcp[30000:110000] = 30                                  # one run of 80001 positions
    da[30000] = 0; da[30001] = 60                            # a read and a genome inside it
    cl, nc, ml = ctx.detect(lcp, da, 50, 16)
    ecl, enc, eml = O.detect(lcp, da, 50, 16)
    assert (nc, ml) == (enc, eml) and np.array_equal(cl, ecl) and ml > 80000


def test_docid_out_of_range_is_reported(ctx):
    import lime_amd
    lcp = np.array([0, 20, 20, 0], np.uint32)
    da = np.array([0, 9, 1, 0], np.uint32)              # 9 >= n_reads + n_refs
    with pytest.raises(lime_amd.LimeError) as e:
        ctx.fused(lcp, da, None, 1, 2, 16)
    assert e.value.code == -6


# ---- device-resident API: synthetic generator, shards ----------------------------------------
def test_device_synth_equals_oracle(ctx):
    import torch
    n = 100000
    for mode in (0, 1):
        l = torch.empty(n, dtype=torch.int32, device="cuda")
        d = torch.empty(n, dtype=torch.int32, device="cuda")
        e = torch.empty(n, dtype=torch.uint8, device="cuda")
        ctx.synth_dev(42, 5000, n, 321, 17, 16, mode, l, d, e)
        torch.cuda.synchronize()
        ol, od, oe = O.synth(42, 5000, n, 321, 17, 16, mode)
        assert np.array_equal(l.cpu().numpy().view(np.uint32), ol)
        assert np.array_equal(d.cpu().numpy().view(np.uint32), od)
        assert np.array_equal(e.cpu().numpy(), oe)


@pytest.mark.parametrize("n_shards", [2, 3, 5])
def test_position_range_shards_sum_to_whole(ctx, n_shards):
    """shard by contiguous tile-aligned ranges with a halo; per-shard tables add up (mod 256)
    to the single-pass table and the counters combine (sum / max) -- the multi-GPU scheme."""
    import torch
    from lime_amd.dist import shard_ranges
    n, nr, ng = 700001, 200, 13
    lcp, da, eb = O.synth(9, 0, n, nr, ng, 16, 1)
    lcp[300000:300900] = 20                      # a run across a shard cut region
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    exp = O.score(da, eb, cl, nr, ng, threads=4)
    total = np.zeros((nr, ng), np.uint8)
    tot_c, tot_m = 0, 0
    for lo, hi, hi_halo in shard_ranges(n, n_shards, halo=65536 + 4096):
        tl = torch.from_numpy(lcp[lo:hi_halo].view(np.int32)).cuda()
        td = torch.from_numpy(da[lo:hi_halo].view(np.int32)).cuda()
        te = torch.from_numpy(eb[lo:hi_halo]).cuda()
        sim = torch.zeros(lime_sim_bytes(nr, ng), dtype=torch.uint8, device="cuda")
        ctx.fused_dev(tl, td, te, hi - lo, hi_halo - lo, hi_halo == n, nr, ng, 16, sim)
        s, rc = ctx.stats()
        assert rc == 0
        tot_c += s.n_clusters; tot_m = max(tot_m, s.max_len)
        total = (total + sim[:nr * ng].cpu().numpy().reshape(nr, ng)).astype(np.uint8)
    assert (tot_c, tot_m) == (nc, ml)
    assert np.array_equal(total, exp)


@pytest.mark.parametrize("shape", ["C2", "C3", "C4", "C5"])
def test_full_size_paths_agree(ctx, shape):
    """BASELINE.json's full sizes (configs[1]: 10^8 symbols, 10^5 x 500, EBWT=1; configs[2]: 10^9 symbols,
    10^6 x 5000, EBWT=0), inputs generated on the device.  Too big for the oracle, so size-independent
    properties: (1) the table of one fused pass == the table accumulated over 3 position-range shards with
    halos (ownership + mod-256 addition); (2) == the table of the two-program flow, detection then scoring of
    the emitted cluster list (independent kernels); (3) the cluster list is sorted, its length and longest
    record equal the fused counters; (4) the row maxima of k_choose == a plain reduction of the table."""
    import ctypes as C
    import torch
    import lime_amd
    lime_amd.trim_cache()                        # (blocks that closed contexts left with the library: these shapes want the whole device)
    from lime_amd.dist import shard_ranges
    # C4: the shapes of configs[3] (setB2: 20 249 373 reads x 930 genomes = 18.8 GB table, README.md:137) on 2*10^9
    # synthetic symbols, EBWT=1; its four position-range shards run one after the other on this one GPU
    # C5: the shapes of configs[4] (SRR1804065 fwd + rc over the full reference database, Datasets/README.md:67: about 10^10
    # symbols, 3 423 genomes; 3 * 10^6 reads -> a 10.3 GB table), EBWT=1: 90 GB of arrays resident on the one GPU; its eight
    # position-range shards run one after the other, their edge words are combined like the ranks' (lime_combine_edges)
    n, nr, ng, ebwt_on = {"C2": (100_000_000, 100_000, 500, True), "C3": (1_000_000_000, 1_000_000, 5000, False),
                          "C4": (2_000_000_000, 20_249_373, 930, True), "C5": (10_000_000_000, 3_000_000, 3423, True)}[shape]
    n_shards = {"C4": 4, "C5": 8}.get(shape, 3)
    alpha, dev = 16, torch.device("cuda:0")
    lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
    eb = torch.empty(n, dtype=torch.uint8, device=dev) if ebwt_on else None
    ctx.synth_dev(42, 0, n, nr, ng, alpha, 0, lcp, da, eb)
    tb = lime_amd.sim_bytes(nr, ng)
    A = torch.empty(tb, dtype=torch.uint8, device=dev)
    ctx.fused_dev(lcp, da, eb, n, n, True, nr, ng, alpha, A, True)
    sA, rc = ctx.stats(); assert rc == 0
    assert sA.n_clusters > n // 30 and sA.n_updates > 0
    # (1) three (C4: four) shards into one table
    B = torch.empty(tb, dtype=torch.uint8, device=dev)
    tot_c, tot_m, tot_u, edges = 0, 0, 0, []
    for k, (lo, hi, hh) in enumerate(shard_ranges(n, n_shards)):
        ctx.fused_dev(lcp[lo:], da[lo:], None if eb is None else eb[lo:], hi - lo, hh - lo, hh == n, nr, ng, alpha, B, k == 0)
        s, rc = ctx.stats(); assert rc == 0
        tot_c += s.n_clusters; tot_m = max(tot_m, s.max_len); tot_u += s.n_updates; edges.append(s.edge)
    from lime_amd.dist import combine_edges
    combine_edges(edges)                                 # no cluster among runs that cross shard borders beyond the halo
    assert (tot_c, tot_m, tot_u) == (sA.n_clusters, sA.max_len, sA.n_updates)
    assert torch.equal(A, B)
    del B
    if shape == "C5":                                    # (2), (3) would need 18 GB of records on top: the row scan only
        mx = torch.empty(nr, dtype=torch.uint8, device=dev); nz = torch.empty(nr, dtype=torch.int32, device=dev)
        ctx.choose_dev(A, nr, ng, mx, nz)
        t2 = A[:nr * ng].view(nr, ng)
        assert torch.equal(mx, t2.amax(dim=1))
        assert int(nz.sum()) == int(torch.count_nonzero(t2))
        return
    # (2),(3) detection, then scoring of the list
    ptr, nc, ml = ctx.detect_dev(lcp, da, n, n, True, 0, nr, alpha)
    assert (nc, ml) == (sA.n_clusters, sA.max_len)
    rec = torch.empty((nc, 2), dtype=torch.int64, device=dev)
    assert lime_amd._lib.hip_memcpy_d2d(rec.data_ptr(), ptr, nc * 16) == 0
    assert bool((rec[1:, 0] > rec[:-1, 0]).all()) and int(rec[:, 1].min()) >= 2 and int(rec[:, 1].max()) == ml
    assert bool((rec[:-1, 0] + rec[:-1, 1] <= rec[1:, 0]).all())          # clusters do not overlap
    Cc = torch.empty(tb, dtype=torch.uint8, device=dev)
    ctx.score_dev(da, eb, n, rec.data_ptr(), nc, nr, ng, Cc, True)
    s, rc = ctx.stats(); assert rc == 0 and s.n_updates == sA.n_updates
    assert torch.equal(A, Cc)
    del Cc, rec
    # (4) row scan
    mx = torch.empty(nr, dtype=torch.uint8, device=dev); nz = torch.empty(nr, dtype=torch.int32, device=dev)
    ctx.choose_dev(A, nr, ng, mx, nz)
    t2 = A[:nr * ng].view(nr, ng)
    assert torch.equal(mx, t2.amax(dim=1))
    assert int(nz.sum()) == int(torch.count_nonzero(t2))


@pytest.mark.parametrize("nr,ng,mode", [(1, 1, 0), (2, 2, 1), (3, 5, 1), (2, 40, 0)])
def test_repeated_documents_everywhere(ctx, nr, ng, mode):
    """very few documents: nearly every cluster holds a repeated document, far more per window than a wave's
    LDS store takes at once (it is flushed again and again inside a scoring round); counts wrap modulo 256
    many times"""
    n = 400003
    lcp, da, eb = O.synth(1234 + nr, 0, n, nr, ng, 16, mode)
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    for e in (eb, None):
        exp = O.score(da, e, cl, nr, ng, threads=4)
        sim, gnc, gml = ctx.fused(lcp, da, e, nr, ng, 16)
        assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp)
        assert np.array_equal(ctx.score(da, e, cl, nr, ng), exp)


@pytest.mark.parametrize("layout", ["blocks", "interleaved", "genomes_first"])
def test_pair_score_of_32_in_a_64_symbol_cluster(ctx, layout):
    """one cluster of 64 symbols holding ONE read x32 and ONE genome x32: t = 32, the largest score a cluster
    scored inside the scan can give (ClusterBWT_DA.cpp:232-250 adds 32; a 5-bit field would drop it)"""
    n = 3000
    lcp = np.zeros(n, np.uint32); da = np.full(n, 1, np.uint32)          # filler: genome-only, no runs
    s0 = 1000
    lcp[s0 + 1:s0 + 64] = 20
    docs = {"blocks": [0] * 32 + [1] * 32, "interleaved": [0, 1] * 32, "genomes_first": [1] * 32 + [0] * 32}[layout]
    da[s0:s0 + 64] = docs
    eb = np.full(n, ord("A"), np.uint8)                                  # EBWT=1: one repeated base
    cl, nc, ml = O.detect(lcp, da, 1, 16)
    assert nc == 1 and ml == 64
    for e in (eb, None):
        exp = O.score(da, e, cl, 1, 1)
        assert int(exp[0, 0]) == 32
        sim, gnc, gml = ctx.fused(lcp, da, e, 1, 1, 16)
        assert (gnc, gml) == (1, 64) and np.array_equal(sim, exp)
        sim, gnc, gml = ctx.fused_stream(lcp, da, e, 1, 1, 16, chunk=4096)
        assert (gnc, gml) == (1, 64) and np.array_equal(sim, exp)
        assert np.array_equal(ctx.score(da, e, cl, 1, 1), exp)


@pytest.mark.parametrize("length", [33, 48, 60, 61, 62, 63, 64, 65, 66])
@pytest.mark.parametrize("nr,ng", [(1, 1), (2, 1), (2, 3)])
def test_clusters_of_up_to_64_symbols_with_few_documents(ctx, length, nr, ng):
    """clusters just below / at / above the 64-symbol in-scan limit with very few documents (large pair scores),
    at many window offsets, both builds"""
    rng = np.random.default_rng(length * 131 + nr * 7 + ng)
    n = 60000
    lcp = np.zeros(n, np.uint32)
    for s0 in range(50, n - 200, 173):                                  # cluster starts at every window phase
        lcp[s0 + 1:s0 + length] = 17
    da = np.where(rng.random(n) < 0.5, rng.integers(0, nr, n), nr + rng.integers(0, ng, n)).astype(np.uint32)
    eb = rng.choice(np.frombuffer(b"AACN", np.uint8), n).astype(np.uint8)
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    assert ml == length
    for e in (eb, None):
        exp = O.score(da, e, cl, nr, ng)
        sim, gnc, gml = ctx.fused(lcp, da, e, nr, ng, 16)
        assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp)
        assert np.array_equal(ctx.score(da, e, cl, nr, ng), exp)


def _long_run_case(kind):
    """700 000 positions with one run of 250 000 (far longer than the 69 632-position halo of a chunk / shard):
    genome-only, read-only, or reads with one genome near its end (a cluster of 250 000: refused)"""
    n, nr, ng = 700000, 40, 9
    lcp, da, eb = O.synth(99, 0, n, nr, ng, 16, 0)
    a, b = 200000, 450000
    lcp[a + 1:b] = 40; lcp[a] = 0; lcp[b] = 0
    if kind == "genomes":
        da[a:b] = nr + (np.arange(b - a) % ng)
    elif kind == "reads":
        da[a:b] = np.arange(b - a) % nr
    else:
        da[a:b] = np.arange(b - a) % nr
        da[b - 10] = nr + 2                             # reads for 249 990 positions, then one genome: a cluster after all
    return n, nr, ng, lcp.astype(np.uint32), da.astype(np.uint32), eb


@pytest.mark.parametrize("kind", ["genomes", "reads"])
def test_long_run_that_is_no_cluster_crosses_chunks_and_shards(ctx, kind):
    """the reference reads on without limit and refuses only CLUSTERS longer than 65536 (ClusterLCP.cpp:246-264,
    ClusterBWT_DA.cpp:558-562): a genome-only (read-only) run of 250 000 positions across chunk / shard borders is
    nothing at all"""
    import torch
    import lime_amd
    from lime_amd.dist import shard_ranges, combine_edges
    n, nr, ng, lcp, da, eb = _long_run_case(kind)
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    assert ml < 100
    exp = O.score(da, eb, cl, nr, ng, threads=4)
    for chunk in (4096, 65536):
        sim, gnc, gml = ctx.fused_stream(lcp, da, eb, nr, ng, 16, chunk=chunk)
        assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp)
    total = np.zeros((nr, ng), np.uint8); tot_c = 0; edges = []; saw_open = False
    for lo, hi, hh in shard_ranges(n, 5):
        tl = torch.from_numpy(lcp[lo:hh].view(np.int32)).cuda(); td = torch.from_numpy(da[lo:hh].view(np.int32)).cuda()
        te = torch.from_numpy(eb[lo:hh]).cuda()
        sim = torch.zeros(lime_sim_bytes(nr, ng), dtype=torch.uint8, device="cuda")
        ctx.fused_dev(tl, td, te, hi - lo, hh - lo, hh == n, nr, ng, 16, sim)
        s, rc = ctx.stats()
        assert rc == 0 or (rc == -5 and s.edge & 8), (rc, s.edge)          # LIME_ERR_HALO only with an open run reported
        saw_open |= bool(s.edge & 8)
        edges.append(s.edge); tot_c += s.n_clusters
        total = (total + sim[:nr * ng].cpu().numpy().reshape(nr, ng)).astype(np.uint8)
    assert saw_open
    combine_edges(edges)                                                   # no cluster among the border-crossing runs
    assert tot_c == nc and np.array_equal(total, exp)


def test_long_cluster_across_chunks_is_refused(ctx):
    import torch
    import lime_amd
    from lime_amd.dist import shard_ranges, combine_edges
    n, nr, ng, lcp, da, eb = _long_run_case("cluster")
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    assert ml == 250000
    with pytest.raises(lime_amd.LimeError) as e:
        ctx.fused_stream(lcp, da, eb, nr, ng, 16, chunk=65536)
    assert e.value.code == -4
    edges = []
    for lo, hi, hh in shard_ranges(n, 5):
        tl = torch.from_numpy(lcp[lo:hh].view(np.int32)).cuda(); td = torch.from_numpy(da[lo:hh].view(np.int32)).cuda()
        sim = torch.zeros(lime_sim_bytes(nr, ng), dtype=torch.uint8, device="cuda")
        ctx.fused_dev(tl, td, None, hi - lo, hh - lo, hh == n, nr, ng, 16, sim)
        s, rc = ctx.stats()
        assert rc == 0 or (rc == -5 and s.edge & 8)     # no shard sees a read and a genome of the run by itself
        edges.append(s.edge)
    with pytest.raises(lime_amd.LimeError) as e:
        combine_edges(edges)
    assert e.value.code == -4


@pytest.mark.parametrize("n", [1024, 4096, 5000, 65536, 300032])
def test_nothing_beyond_n_avail_is_used(ctx, n):
    """the arrays handed to lime_fused_dev / lime_detect_dev end exactly at n_avail inside larger device buffers whose
    continuation is poison (runs that would go on, reads and genomes that would complete clusters): results must be
    those of the first n positions alone -- n a multiple of the 1024-position window (no partial window, nothing
    after the last one but the poison) and not"""
    import torch
    nr, ng = 300, 25
    lcp, da, eb = O.synth(4242 + n, 0, n + 2048, nr, ng, 16, 1)
    lcp[n:] = 99                                   # the data "goes on" as one long run of alternating read / genome
    da[n:] = np.where(np.arange(2048) % 2 == 0, 0, nr).astype(np.uint32)
    cl, nc, ml = O.detect(lcp[:n], da[:n], nr, 16)
    tl = torch.from_numpy(lcp.view(np.int32)).cuda(); td = torch.from_numpy(da.view(np.int32)).cuda(); te = torch.from_numpy(eb).cuda()
    for e, et in ((eb, te), (None, None)):
        exp = O.score(da[:n], None if e is None else e[:n], cl, nr, ng, threads=4)
        sim = torch.zeros(lime_sim_bytes(nr, ng), dtype=torch.uint8, device="cuda")
        ctx.fused_dev(tl, td, et, n, n, True, nr, ng, 16, sim)
        s, rc = ctx.stats()
        assert rc == 0 and (s.n_clusters, s.max_len) == (nc, ml)
        assert np.array_equal(sim[:nr * ng].cpu().numpy().reshape(nr, ng), exp)
    ptr, gnc, gml = ctx.detect_dev(tl, td, n, n, True, 0, nr, 16)
    assert (gnc, gml) == (nc, ml)


def lime_sim_bytes(nr, ng):
    import lime_amd
    return lime_amd.sim_bytes(nr, ng)


@pytest.mark.parametrize("max_blocks", [1, 3, 7, 64])
def test_persistent_workgroups_walk_many_tiles(max_blocks, monkeypatch):
    """few persistent workgroups, each walking many tiles with the next tile prefetched."""
    import lime_amd
    monkeypatch.setenv("LIME_MAX_BLOCKS", str(max_blocks))
    c = lime_amd.Context()
    try:
        for seed, mode in ((5, 0), (6, 1)):
            n, nr, ng = 700000 + seed, 300, 20
            lcp, da, eb = O.synth(seed, 0, n, nr, ng, 16, mode)
            cl, nc, ml = O.detect(lcp, da, nr, 16)
            gcl, gnc, gml = c.detect(lcp, da, nr, 16)
            assert np.array_equal(gcl, cl) and (gnc, gml) == (nc, ml)
            for e in (eb, None):
                exp = O.score(da, e, cl, nr, ng, threads=4)
                sim, gnc, gml = c.fused(lcp, da, e, nr, ng, 16)
                assert (gnc, gml) == (nc, ml)
                assert np.array_equal(sim, exp)
    finally:
        c.close()


@pytest.mark.parametrize("period", [2, 3, 4, 5])
@pytest.mark.parametrize("n", [250, 256, 300, 512, 1030, 5000])
def test_dense_small_clusters(ctx, period, n):
    """more than 64 small clusters per 512-position window (several hand-out rounds per window),
    ends of data at every offset of a window"""
    rng = np.random.default_rng(period * 10007 + n)
    nr = ng = 3000
    lcp = np.full(n, 20, np.uint32); lcp[::period] = 0
    da = np.where(rng.random(n) < 0.5, rng.integers(0, nr, n), nr + rng.integers(0, ng, n)).astype(np.uint32)
    eb = rng.choice(np.frombuffer(b"ACGTN", np.uint8), n).astype(np.uint8)
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    for e in (eb, None):
        exp = O.score(da, e, cl, nr, ng)
        sim, gnc, gml = ctx.fused(lcp, da, e, nr, ng, 16)
        assert (gnc, gml) == (nc, ml)
        assert np.array_equal(sim, exp)


@pytest.mark.parametrize("dense_min", ["0", "3", "64", "4294967295"])
@pytest.mark.parametrize("path", ["cas", "bin"])
def test_dense_windows_list_their_two_symbol_clusters_apart(monkeypatch, golden, dense_min, path):
    """Round 6: a window with more than dense_min accepted clusters (64) splits its cluster list -- the 2-symbol clusters, found by bit arithmetic
    on the chunks' head masks with the next chunk's / the read-ahead's first two bits, behind the others -- and scores them in rounds of their
    own.  Every golden vector of the reference with the split in EVERY window (0), from 4 clusters on, at the default and never, both update
    paths, both builds; then runs of period 2 .. 5 (2-symbol clusters back to back, up to 512 per window) whose data ends at every offset of
    a window and of a 16-position chunk, cut into position-range shards at places inside such runs (ownership of the cluster at the cut)."""
    import torch
    import lime_amd
    monkeypatch.setenv("LIME_UPDATE_PATH", path)
    c = lime_amd.Context()
    try:
        c.set_option("dense_min", dense_min)
        g = golden
        for ebwt_on, key in ((True, "sim_e1"), (False, "sim_e0")):
            sim, nc, ml = c.fused(g["lcp"], g["da"], g["ebwt"] if ebwt_on else None, g["n_reads"], g["n_refs"], g["alpha"])
            assert nc == len(g["clrs"]) and np.array_equal(sim, g[key]), (g["name"], dense_min, path, ebwt_on)
        if g["name"] != GOLDEN_FIRST:
            return
        rng = np.random.default_rng(77)
        nr = ng = 2000
        for period in (2, 3, 5):
            for n in (1024, 1025, 1039, 1040, 2047, 2050, 3000 + period):
                lcp = np.full(n, 20, np.uint32); lcp[::period] = 0
                if period == 3:
                    lcp[1::6] = 0                              # 1- and 2-symbol segments alternate: heads at b, b + 1 and b, b + 2 patterns
                da = np.where(rng.random(n) < 0.5, rng.integers(0, nr, n), nr + rng.integers(0, ng, n)).astype(np.uint32)
                eb = rng.choice(np.frombuffer(b"ACGTNR\x00$", np.uint8), n).astype(np.uint8)
                cl, nc, ml = O.detect(lcp, da, nr, 16)
                for e in (eb, None):
                    exp = O.score(da, e, cl, nr, ng)
                    sim, gnc, gml = c.fused(lcp, da, e, nr, ng, 16)
                    assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp), (period, n, dense_min, path, e is None)
                # two shards cut inside the runs: the cluster at the cut belongs to the shard of its head
                tl = torch.from_numpy(lcp.view(np.int32)).cuda(); td = torch.from_numpy(da.view(np.int32)).cuda(); te = torch.from_numpy(eb).cuda()
                exp = O.score(da, eb, cl, nr, ng)
                for cut in (504, 1016, 1024, 1032):           # (multiples of 8: a shard's arrays start 16-byte aligned, ebwt 8)
                    if cut >= n:
                        continue
                    acc = torch.zeros(lime_sim_bytes(nr, ng), dtype=torch.uint8, device="cuda")
                    tot = 0
                    for lo, hi in ((0, cut), (cut, n)):
                        c.fused_dev(tl[lo:], td[lo:], te[lo:], hi - lo, n - lo, True, nr, ng, 16, acc, lo == 0)
                        s, rc = c.stats(); assert rc == 0
                        tot += s.n_clusters
                    assert tot == nc and np.array_equal(acc[:nr * ng].cpu().numpy().reshape(nr, ng), exp), (period, n, cut, dense_min, path)
    finally:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pct", [0, 50, 100])
@pytest.mark.parametrize("path,ebwt_on", [("cas", False), ("bin", False), ("cas", True)])
def test_window_handout_repeats(monkeypatch, pct, path, ebwt_on):
    """The scan's waves draw their windows from counters (LDS inside a workgroup, a device word for the last
    rounds): whatever the split between fixed and handed-out rounds, and however the waves race, every window is
    taken exactly once -- the same pass repeated gives the same counters and table, equal to the all-fixed split's.
    (A workgroup whose two chunk fetches were answered out of order once dropped a chunk near the end of the data.)"""
    import torch
    import lime_amd
    n, nr, ng, alpha = 40_000_000 + 12_345, 200_000, 700, 16
    dev = torch.device("cuda:0")
    lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
    eb = torch.empty(n, dtype=torch.uint8, device=dev) if ebwt_on else None
    monkeypatch.setenv("LIME_UPDATE_PATH", path)
    tb = lime_amd.sim_bytes(nr, ng)
    want = None
    for p in (100, pct):
        monkeypatch.setenv("LIME_SCAN_STATIC_PCT", str(p))
        c = lime_amd.Context()
        try:
            if want is None:
                c.synth_dev(7, 0, n, nr, ng, alpha, 0, lcp, da, eb)
            for _ in range(1 if want is None else 6):
                A = torch.empty(tb, dtype=torch.uint8, device=dev)
                c.fused_dev(lcp, da, eb, n, n, True, nr, ng, alpha, A, True)
                s, rc = c.stats(); assert rc == 0
                got = (s.n_clusters, s.max_len, s.n_updates)
                if want is None:
                    want, ref = got, A
                else:
                    assert got == want
                    assert torch.equal(A, ref)
        finally:
            c.close()
