#!/usr/bin/env python3
"""tools/fuzz.py [seconds] [seed] -- random inputs through every entry point of the hot path against the oracle (bit-exact), on
the GPU box.  Shapes, generator parameters, cluster-length regimes, update paths, chunk sizes and shard counts are
drawn at random; stops at the first difference with a description that reproduces it.  Not part of the test suite
(time-boxed soak); the fixed cases it found nothing beyond are in tests/."""
import os, sys, time
os.environ["LIME_TEST_HOOKS"] = "1"           # the library reads its LIME_<KNOB> variables only in a process that says it is a test (lime_init)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lime_amd
from lime_amd.dist import shard_ranges, combine_edges
from oracle import oracle_py as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
it = 0
t_print = time.time()
stats = {"cases": 0, "symbols": 0, "binned": 0, "shards": 0, "streams": 0, "p64": 0, "choose_free": 0}


def expected_choose(sim, norm, beta):
    mx = sim.max(axis=1) if sim.shape[1] else np.zeros(sim.shape[0], np.uint8)
    ok = (mx.astype(np.float32) / np.float32(norm)) > np.float32(beta)
    off = np.zeros(sim.shape[0] + 1, np.uint64); rows = []
    for r in range(sim.shape[0]):
        nzc = np.nonzero(sim[r])[0] if ok[r] else np.zeros(0, np.int64)
        if len(nzc): rows.append(np.stack([nzc.astype(np.uint32), sim[r][nzc].astype(np.uint32)], axis=1))
        off[r + 1] = off[r] + len(nzc)
    return mx, off, (np.concatenate(rows) if rows else np.zeros((0, 2), np.uint32))

while time.time() < t_end:
    it += 1
    rng = np.random.default_rng(seed0 * 100003 + it)
    n = int(rng.choice([rng.integers(1, 3000), rng.integers(3000, 200000), rng.integers(200000, 1500000)]))
    nr = int(rng.choice([1, 2, rng.integers(3, 50), rng.integers(50, 5000), rng.integers(5000, 200000)]))
    ng = int(rng.choice([1, 2, rng.integers(3, 50), rng.integers(50, 3000)]))
    alpha = int(rng.choice([1, 2, 16, 16, 16, 30]))
    mode = int(rng.integers(0, 2))
    lcp, da, eb = O.synth(int(rng.integers(1, 1 << 30)), int(rng.integers(0, 1 << 20)), n, nr, ng, alpha, mode)
    regime = int(rng.integers(0, 5))
    if regime == 1:                      # long runs
        k = int(rng.integers(1, 6))
        for _ in range(k):
            a0 = int(rng.integers(0, n)); ln = int(rng.integers(min(10, n), min(n, 70000) + 1))
            lcp[a0:a0 + ln] = alpha + 3
    elif regime == 2:                    # dense short clusters
        per = int(rng.integers(2, 7)); lcp[:] = alpha; lcp[::per] = 0
    elif regime == 3 and n > 10:         # few documents: repeats everywhere
        da = np.where(da < nr, da % min(nr, 2), nr + (da - nr) % min(ng, 2)).astype(np.uint32)
    lcp[0] = int(rng.choice([0, alpha]))  # leading positions before the first head are ignored
    lcp = lcp.astype(np.uint32)
    if rng.random() < 0.2:
        eb = rng.choice(np.frombuffer(b"ACGTNRYKMSWBDHV\x00$#acgt", np.uint8), n).astype(np.uint8)
    cl, nc, ml = O.detect(lcp, da, nr, alpha)
    desc = f"it={it} seed0={seed0} n={n} nr={nr} ng={ng} alpha={alpha} mode={mode} regime={regime}"
    path = "bin" if rng.random() < 0.5 else "cas"
    os.environ["LIME_UPDATE_PATH"] = path
    if path == "bin" and rng.random() < 0.5:
        os.environ["LIME_BIN_LEVELS"] = f"{int(rng.integers(1, 30))},{int(rng.integers(1, 200))}"
    else:
        os.environ.pop("LIME_BIN_LEVELS", None)
    # round 5: the 64-bit-position partition kernels, the record positions starting at a random base around multiples of 2^32
    do_choose = path == "bin" and rng.random() < 0.4 and nr <= 20000 and nr * ng > 70000
    if do_choose: os.environ["LIME_BIN_LEVELS"] = f"1,{int(rng.integers(1, 9))}"        # (a second level wherever the table has two regions)
    os.environ.pop("LIME_P64_TEST_BASE", None); os.environ.pop("LIME_FORCE_P64", None)
    p64 = path == "bin" and rng.random() < 0.5
    if p64:
        os.environ["LIME_FORCE_P64"] = "1"
        if rng.random() < 0.7:
            os.environ["LIME_P64_TEST_BASE"] = str(int(rng.integers(1, 5)) * (1 << 32) - int(rng.integers(0, 2 * n + 64)))
    ctx = lime_amd.Context()
    dense_min = str(rng.choice([0, 3, 64, 64, 4294967295]))           # round 6: from how many clusters on a window lists its 2-symbol clusters apart
    ctx.set_option("dense_min", dense_min)
    no_direct = str(int(rng.random() < 0.3))                          # round 6: records through the update queue instead of written by the scorers
    ctx.set_option("no_direct", no_direct)
    desc += f" dense_min={dense_min} no_direct={no_direct}"
    try:
        gcl, gnc, gml = ctx.detect(lcp, da, nr, alpha)
        assert (gnc, gml) == (nc, ml) and np.array_equal(gcl, cl), "detect: " + desc
        for e in (eb, None):
            tag = desc + f" path={path} levels={os.environ.get('LIME_BIN_LEVELS')} ebwt={e is not None}"
            if ml > 65536:
                try:
                    ctx.fused(lcp, da, e, nr, ng, alpha); raise AssertionError("no MAXLEN error: " + tag)
                except lime_amd.LimeError as ex:
                    assert ex.code == -4, tag
                continue
            exp = O.score(da, e, cl, nr, ng, threads=8)
            sim, gnc, gml = ctx.fused(lcp, da, e, nr, ng, alpha)
            assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp), "fused: " + tag
            r = rng.random()
            if r < 0.35:
                chunk = int(rng.choice([4096, 8192, 65536, 262144]))
                sim, gnc, gml = ctx.fused_stream(lcp, da, e, nr, ng, alpha, chunk=chunk)
                assert (gnc, gml) == (nc, ml) and np.array_equal(sim, exp), f"stream chunk={chunk}: " + tag
                stats["streams"] += 1
            elif r < 0.6:
                assert np.array_equal(ctx.score(da, e, cl[rng.permutation(len(cl))] if len(cl) else cl, nr, ng), exp), "score: " + tag
            elif r < 0.85 and n > 8192:
                ns = int(rng.integers(2, 6)); tot = np.zeros((nr, ng), np.uint8); tc = 0; edges = []
                for lo, hi, hh in shard_ranges(n, ns):
                    tl = torch.from_numpy(lcp[lo:hh].view(np.int32)).cuda(); td = torch.from_numpy(da[lo:hh].view(np.int32)).cuda()
                    te = None if e is None else torch.from_numpy(e[lo:hh]).cuda()
                    st = torch.full((lime_amd.sim_bytes(nr, ng),), 3, dtype=torch.uint8, device="cuda")
                    ctx.fused_dev(tl, td, te, hi - lo, hh - lo, hh == n, nr, ng, alpha, st, True)
                    s, rc = ctx.stats()
                    assert rc == 0 or (rc == -5 and s.edge & 8), f"shard rc={rc}: " + tag
                    edges.append(s.edge); tc += s.n_clusters
                    tot = (tot + st[:nr * ng].cpu().numpy().reshape(nr, ng)).astype(np.uint8)
                combine_edges(edges)
                assert tc == nc and np.array_equal(tot, exp), f"shards={ns}: " + tag
                stats["shards"] += 1
            if do_choose:                                                      # clusterChoose without the table (forced wherever the layout has a second level)
                norm, beta = 85, float(rng.choice([0.0, 0.012, 0.03, 0.3]))
                ctx.set_option("choose_free", "1"); ctx.set_option("apply_wide", str(int(rng.integers(0, 2))))
                tl = torch.from_numpy(lcp.view(np.int32)).cuda(); td = torch.from_numpy(da.view(np.int32)).cuda()
                te = None if e is None else torch.from_numpy(e).cuda()
                mx, off, prs, st = ctx.fused_choose_dev(tl, td, te, n, nr, ng, alpha, norm, beta)
                emx, eoff, eprs = expected_choose(exp, norm, beta)
                assert np.array_equal(mx, emx) and np.array_equal(off, eoff) and np.array_equal(prs, eprs), f"fused_choose beta={beta}: " + tag
                ctx.set_option("choose_free", ""); ctx.set_option("apply_wide", "")
                stats["choose_free"] += ctx.host_times()["choose_without_table"] > 0
            s, rc = ctx.stats()
        stats["cases"] += 1; stats["symbols"] += n; stats["binned"] += path == "bin"; stats["p64"] += p64
        if time.time() - t_print > 30:                   # a line now and then: a silent GPU run is taken for hung
            print("fuzz ...", stats, flush=True); t_print = time.time()
    finally:
        ctx.close()
print("fuzz ok:", stats, "in", round(budget), "s")
