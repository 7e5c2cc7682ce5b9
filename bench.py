#!/usr/bin/env python3
"""bench.py -- throughput of the LiME hot path (ClusterLCP + ClusterBWT_DA) on MI355X.

Metric (BASELINE.json): eBWT symbols/s processed by detect+score, arrays resident in HBM.

N=1 (default `python bench.py`): BASELINE.json configs[2], the largest single-GPU configuration -- synthetic S of
10^9 symbols, 10^6 reads x 5000 genomes (5 GB table), alpha=16, EBWT=0 (8 B/symbol), generator of SURVEY.md 8d
(seed 42, identical on CPU and GPU).  A step = one fused pass: scan + score + the finished score table (cleared or
rebuilt every step).  The same JSON line carries, under "also", configs[1] (10^8, 10^5 x 500, EBWT=1), the
"clustered" generator on that shape, and an N = 10^10 single-GPU pass (10^6 x 1000 table) -- the one-GPU point
of the scaling series.

N>1 (launched by torch.distributed.run, one rank per GPU): STRONG scaling of the north_star series -- a fixed
collection of 10^10 symbols (10^6 reads x 1000 genomes, EBWT=0) cut into contiguous tile-aligned position ranges
with a read-ahead halo; a step = every rank's pass over its range + the ONE exchange of the path, a reduce-scatter of
the per-rank uint8 tables (sum modulo 256, by read-row blocks) issued through the C ABI (lime_comm_*: RCCL
ncclReduceScatter(ncclUint8, ncclSum)); the step ends when the exchange has completed (exposed -- what a single pass
pays).  The same invocation repeats the series three more times: `also.overlapped` (the exchange of step k under the scan of
step k+1, two table buffers -- what the four passes of LiME_paired.sh allow), `also.sparse_exchange` (owner-partitioned
exchange of the update records instead of whole tables) and `also.auto` (whichever of the two a probe pass says moves fewer
bytes); every entry carries the slowest rank's parts of a pass and the exchange's own time, and the line carries RCCL's
own rank count (ncclCommCount) and the result of the start-up check that its uint8 sum wraps.  `--scaling weak` keeps
configs[2] per GPU.

Prints ONE JSON line on rank 0 (the driver's contract) with two extra objects:
  roofline     HBM bound: algorithmic bytes (8 or 9 B/symbol x symbols per launch) / average duration of the scan
               kernel k_scan measured with HIP events on its stream; also the whole pass
  cpu_baseline the reference's own OpenMP programs (oracle/_ref, kind "reference") on a bounded sample of the
               same workload, all host cores and one thread, with the reference's own timer lines
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALPHA, SEED = 16, 42
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
WORKLOADS = {
    # name: symbols, reads, genomes, EBWT, generator mode, description
    "c3": dict(n=1_000_000_000, nr=1_000_000, ng=5000, ebwt=0, mode=0, what="BASELINE.json configs[2]"),
    "c2": dict(n=100_000_000, nr=100_000, ng=500, ebwt=1, mode=0, what="BASELINE.json configs[1]"),
    "c2_clustered": dict(n=100_000_000, nr=100_000, ng=500, ebwt=1, mode=1, what="configs[1] shape, clustered generator (SURVEY 8d)"),
    "n1e10": dict(n=10_000_000_000, nr=1_000_000, ng=1000, ebwt=0, mode=0, what="north_star scaling series, N = 10^10"),
    # the reference's default build (EBWT=1) at the shapes of configs[3] and configs[4], on one GPU (synthetic generator: the real collections are not in the image)
    "c4_shape": dict(n=2_000_000_000, nr=20_249_373, ng=930, ebwt=1, mode=0, what="shape of BASELINE.json configs[3] (setB2: 20 249 373 reads x 930 genomes), 2*10^9 symbols on one GPU"),
    "c5_shape": dict(n=10_000_000_000, nr=3_000_000, ng=3423, ebwt=1, mode=0, what="shape of BASELINE.json configs[4] (3*10^6 reads x 3423 genomes), 10^10 symbols on one GPU"),
    # the same two N = 10^10 shapes with the "clustered" generator (more and longer clusters, 25 % reads): several times the update
    # density of the iid generator -- the regime of real collections, where rounds 1-4 fell back to compare-and-swap (32-bit record positions)
    "c5_clustered": dict(n=10_000_000_000, nr=3_000_000, ng=3423, ebwt=1, mode=1, what="shape of BASELINE.json configs[4], clustered generator, 10^10 symbols on one GPU"),
    "n1e10_clustered": dict(n=10_000_000_000, nr=1_000_000, ng=1000, ebwt=0, mode=1, what="north_star scaling series N = 10^10, clustered generator"),
    # text-derived statistics: tests/golden/text_example.npz (2000 example reads x 3 surrogate genomes, 442 003 symbols, 49.5 % of
    # them in clusters, 0.24 table updates per symbol) laid side by side 226 times, every copy with its own reads and genomes
    "text_tiled": dict(n=226 * 442_003, nr=226 * 2000, ng=226 * 3, ebwt=1, mode=-1, tiled=226,
                       what="tests/golden/text_example.npz x 226 copies (real-text cluster statistics; stands in for configs[0])"),
    # the same arrays with the read ids of all copies spread over the table's rows (r -> 48271 r mod numReads, a bijection): in text_tiled copy c's
    # reads are the rows 2000 c .. 2000 c + 1999, so the records of neighbouring windows all fall into one or two of the table's 1 MB bins -- a
    # locality that real collections do not have (read ids follow the input files' order, not the suffix array's) and that the partition kernels pay
    # for with LDS adds on one word ("Round 5")
    "text_spread": dict(n=226 * 442_003, nr=226 * 2000, ng=226 * 3, ebwt=1, mode=-1, tiled=226, spread=48271,
                        what="text_tiled with the copies' read ids spread over all rows (same clusters, same update density)"),
}


# the secondary workloads of a default run, the largest arrays and record pools first (what they allocate is reused by everything after them)
ALSO_ORDER = ("c5_clustered", "n1e10_clustered", "c5_shape", "n1e10", "c4_shape", "c2", "c2_clustered", "text_tiled", "text_spread")


def describe(wl, n_total, world):
    return ((f"synthetic S (seed {SEED}, generator mode {wl['mode']})" if not wl.get("tiled") else "text-derived S") +
            f": {n_total} symbols, {wl['nr']} reads x {wl['ng']} genomes "
            f"({wl['nr'] * wl['ng'] / 1e9:.2f} GB table), alpha={ALPHA}, EBWT={wl['ebwt']} ({8 + wl['ebwt']} B/symbol) -- {wl['what']}"
            + (f"; cut into {world} position ranges" if world > 1 else ""))


def cpu_baseline(wl, lcp_t, da_t, eb_t, n, sample_n, full=True):
    """The reference's own programs (oracle/_ref) beside the GPU number.  README.md:145 runs them with 4 threads: that point is timed on the
    WHOLE workload when it has at most 10^9 symbols (about 20 s; round 4 timed every point on the first 2*10^8 symbols against the full
    10^6 x 5000 table, which charged the table's allocation and the single-threaded clusterChoose -- about 4.4 s -- to a fifth of the
    symbols); one thread and all cores the box gives this process run on the first `sample_n` symbols (bounded), and for those the
    fixed part (the reference's own `Time:` lines of the table set-up / clusterChoose) is reported separately."""
    import numpy as np
    try:
        aff = sorted(os.sched_getaffinity(0))
    except AttributeError:
        aff = list(range(os.cpu_count() or 1))
    cores = max(1, min(len(aff), 16))          # the pool's sizing rule: 16 cores per GPU
    sample_n = min(sample_n, n)
    ref = os.path.join(ROOT, "oracle", "_ref")
    bwt = "ClusterBWT_DA" if wl["ebwt"] else "ClusterBWT_DA_e0"
    what = f"bench workload (seed {SEED}, mode {wl['mode']}), {wl['nr']}x{wl['ng']}, alpha {ALPHA}, EBWT={wl['ebwt']}"
    if not (os.path.exists(f"{ref}/ClusterLCP") and os.path.exists(f"{ref}/{bwt}")):
        from oracle import oracle_py as O
        lcp = lcp_t[:sample_n].cpu().numpy().view(np.uint32); da = da_t[:sample_n].cpu().numpy().view(np.uint32)
        eb = eb_t[:sample_n].cpu().numpy() if eb_t is not None else None
        t0 = time.perf_counter()
        cl, nc, ml = O.detect(lcp, da, wl["nr"], ALPHA)
        O.score(da, eb, cl, wl["nr"], wl["ng"], threads=cores)
        t1 = time.perf_counter()
        return {"value": sample_n / (t1 - t0), "unit": "symbols/s", "cores": cores, "kind": "port", "sample": f"first {sample_n} symbols of the {what}"}

    def timers(text):
        return [ln.strip() for ln in text.splitlines() if re.match(r"\s*(TIME (clusterAnalyze|clusterChoose)|Time:)", ln)]

    def to_files(base, count):                 # the arrays' first `count` elements as the reference's input files, in pieces (no second copy of 8 GB on the host)
        for t, ext in ((lcp_t, ".lcp"), (da_t, ".da"), (eb_t, ".ebwt")):
            if t is None:
                continue
            with open(base + ext, "wb") as f:
                for lo in range(0, count, 1 << 27):
                    f.write(t[lo:min(count, lo + (1 << 27))].cpu().numpy().tobytes())

    def run(td, base, count, thr):
        t0 = time.perf_counter()
        p1 = subprocess.run([f"{ref}/ClusterLCP", base, str(wl["nr"]), str(wl["ng"]), str(ALPHA), str(thr)], check=True, capture_output=True, cwd=td, timeout=1500)
        t1 = time.perf_counter()
        p2 = subprocess.run([f"{ref}/{bwt}", base, "100", "0.25", str(thr)], check=True, capture_output=True, cwd=td, timeout=1500)
        t2 = time.perf_counter()
        tm = timers(p2.stdout.decode(errors="replace") + p2.stderr.decode(errors="replace"))
        ana = [float(m.group(1)) for ln in tm for m in [re.search(r"TIME clusterAnalyze[^0-9]*([0-9.]+)", ln)] if m]
        out = {"threads": thr, "symbols": count, "symbols_per_s": count / (t2 - t0), "wall_s": {"ClusterLCP": t1 - t0, "ClusterBWT_DA": t2 - t1},
               "reference_timers": {"ClusterLCP": timers(p1.stdout.decode(errors="replace")), "ClusterBWT_DA": tm}}
        if ana:                                # the scan + the per-cluster analysis alone (what the GPU pass replaces), without the table set-up and clusterChoose
            out["symbols_per_s_scan_and_analyze"] = count / ((t1 - t0) + max(ana))
        return out

    runs = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        base = os.path.join(td, "S.fasta")
        do_full = full and n <= 1_000_000_000 and n > sample_n
        to_files(base, n if do_full else sample_n)
        match = None
        if do_full:
            runs["four_threads_whole_workload"] = run(td, base, n, min(4, cores))
            try:                                       # the reference has just written its outputs for the whole workload: compare them
                match = reference_outputs_match(td, base, wl, lcp_t, da_t, eb_t, n, 0.25)
            except Exception as e:
                match = {"all": False, "failed": str(e)}
            for ext in (".lcp", ".da", ".ebwt"):       # cut the files down to the sample for the other two points
                if os.path.exists(base + ext):
                    os.truncate(base + ext, sample_n * (1 if ext == ".ebwt" else 4))
        else:
            runs["four_threads"] = run(td, base, sample_n, min(4, cores))
        runs["all_cores"] = run(td, base, sample_n, cores)
        runs["one_thread"] = run(td, base, sample_n, 1)
    main_run = runs.get("four_threads_whole_workload") or max(runs.values(), key=lambda x: x["symbols_per_s"])
    return {"value": main_run["symbols_per_s"], "unit": "symbols/s", "cores": main_run["threads"], "kind": "reference",
            "sample": (f"the whole {what}: {main_run['symbols']} symbols" if "four_threads_whole_workload" in runs else f"first {sample_n} symbols of the {what}") +
                      "; wall time of the reference's ClusterLCP + ClusterBWT_DA processes (files in page cache; ClusterBWT_DA's wall includes "
                      "allocating/zeroing the table and the single-threaded clusterChoose); the other thread counts ran on the first "
                      f"{sample_n} symbols (runs.*; their fixed table set-up weighs more there: see symbols_per_s_scan_and_analyze)",
            "nproc": os.cpu_count(), "affinity_cpus": len(aff), "runs": runs,
            # the reference's own output files of that whole-workload run, byte for byte against this library on the same arrays (None: no whole-workload run)
            "reference_outputs_match": None if match is None else match["all"], "reference_outputs": match}


def reference_outputs_match(td, base, wl, lcp_t, da_t, eb_t, n, beta):
    """The files the reference's programs have just written for the WHOLE workload (cpu_baseline pays for that run anyway) against this
    library on the same arrays: <base>.out (28 bytes), the .clrs records (the reference's record order is its threads' arrival order,
    ClusterLCP.cpp:229-235: sorted by pStart first), .res.bin / .res.pos through lime_fused_choose_dev + lime_write_res_bin_pairs with and
    without the table.  -> dict of booleans + "all"."""
    import struct
    import numpy as np
    import torch
    import lime_amd
    from lime_amd import _lib
    out = {}
    nr, ng, norm = wl["nr"], wl["ng"], 100 + 1 - ALPHA
    lib = _lib.load()
    c = lime_amd.Context()
    try:
        ptr, nc, ml = c.detect_dev(lcp_t, da_t, n, n, True, 0, nr, ALPHA)
        ref_out = open(os.path.join(td, "S.out"), "rb").read()
        out["out_file"] = struct.pack("<IIIQQ", nr, ng, ALPHA, ml, nc) == ref_out
        ref = np.fromfile(f"{base}.{ALPHA}.clrs", dtype="<u8").reshape(-1, 2)
        ok = len(ref) == nc
        if ok:
            ref = ref[np.argsort(ref[:, 0], kind="stable")]
            rec = torch.empty((nc, 2), dtype=torch.int64, device=lcp_t.device)
            ok = _lib.hip_memcpy_d2d(rec.data_ptr(), ptr, nc * 16) == 0
            for lo in range(0, nc, 1 << 25):
                hi = min(nc, lo + (1 << 25))
                ok = ok and bool(np.array_equal(rec[lo:hi].cpu().numpy().view(np.uint64), ref[lo:hi]))
            del rec
        out["clrs_records"] = bool(ok)
        del ref
        beta32 = float(np.float32(beta))
        for free in ("1", "0"):
            c.set_option("choose_free", free)
            mx, off, pairs, s = c.fused_choose_dev(lcp_t, da_t, eb_t, n, nr, ng, ALPHA, norm, beta32)
            mxp = np.zeros(nr + 1, np.uint8); mxp[:nr] = mx
            offp = np.ascontiguousarray(off, dtype=np.uint64); pb = np.ascontiguousarray(pairs)
            gb, gp = os.path.join(td, "got.bin"), os.path.join(td, "got.pos")
            rc = lib.lime_write_res_bin_pairs(gb.encode(), gp.encode(), mxp.ctypes.data, offp.ctypes.data, pb.ctypes.data if len(pb) else None, nr, norm, beta32)
            same = rc == 0
            for a, b in ((gb, base + ".res.bin"), (gp, base + ".res.pos")):
                same = same and os.path.getsize(a) == os.path.getsize(b) and open(a, "rb").read() == open(b, "rb").read()
            out["res_files_" + ("without_table" if free == "1" else "with_table")] = bool(same)
            del mx, off, pairs, pb
    finally:
        c.close()
    out["all"] = all(out.values())
    return out


def run_pass_series(torch, lime_amd, ldist, wl, n_total, steps, warmup, world, rank, dev, comm, overlap, exchange="dense", options=None):
    """K timed steps of the workload on this rank's position range; returns (seconds, per-step parts, stats, n_own).
    options: lime_set_option knobs for the series' context (the A/B scripts under tools/; bench.py itself passes none)."""
    lo, hi, hi_halo = ldist.shard_ranges(n_total, world)[rank]
    n_own, n_avail = hi - lo, hi_halo - lo
    ctx = lime_amd.Context(dev.index)
    for k_, v_ in (options or {}).items():
        ctx.set_option(k_, v_)
    lcp = torch.empty(n_avail, dtype=torch.int32, device=dev)
    da = torch.empty(n_avail, dtype=torch.int32, device=dev)
    eb = torch.empty(n_avail, dtype=torch.uint8, device=dev) if wl["ebwt"] else None
    sim_bytes = lime_amd.sim_bytes(wl["nr"], wl["ng"])
    blk_bytes = ldist.table_block_bytes(sim_bytes, world)
    nbuf = 2 if (world > 1 and overlap) else 1
    stream0 = torch.cuda.current_stream().cuda_stream
    if world > 1 and exchange == "auto":
        # one pass that stops at the binned records tells how many updates this rank's range makes; the ranks agree on the
        # largest count and take the exchange that moves fewer bytes (lime_amd/dist.py:choose_exchange)
        ctx.synth_dev(SEED, lo, n_avail, wl["nr"], wl["ng"], ALPHA, wl["mode"], lcp, da, eb, stream0)
        ctx.fused_records_dev(lcp, da, eb, n_own, n_avail, hi_halo == n_total, wl["nr"], wl["ng"], ALPHA, stream0)
        s0, rc0 = ctx.stats(stream0)
        if rc0:
            sys.exit(f"scan failed: rc={rc0}")
        exchange = ldist.choose_exchange(comm.max_float(float(s0.n_updates)), sim_bytes)
    sparse = world > 1 and exchange == "sparse"
    if sparse:
        # owner-partitioned exchange: no table per rank, only this rank's block of it (T / world bytes)
        n_bins, bin_shift = ctx.records_layout(wl["nr"], wl["ng"])
        own_block = torch.empty(((n_bins + world - 1) // world) << bin_shift, dtype=torch.uint8, device=dev)
        sims, blks = [None], []
    else:
        sims = [torch.zeros(blk_bytes * world, dtype=torch.uint8, device=dev) for _ in range(nbuf)]
        blks = [torch.empty(blk_bytes, dtype=torch.uint8, device=dev) for _ in range(nbuf)] if world > 1 else []
    stream = torch.cuda.current_stream().cuda_stream
    ex_stream = torch.cuda.Stream(device=dev) if (world > 1 and overlap) else None
    done = [None] * nbuf                       # event: the exchange that read sims[b] has completed
    if wl.get("tiled"):
        import numpy as np
        z = np.load(os.path.join(ROOT, "tests", "golden", "text_example.npz"))
        k, m = wl["tiled"], len(z["lcp"])
        nr1, ng1 = int(z["params"][0]), int(z["params"][1])
        assert world == 1 and n_avail == k * m and wl["nr"] == k * nr1 and wl["ng"] == k * ng1
        l1 = torch.from_numpy(z["lcp"].astype(np.int32)).to(dev); d1 = torch.from_numpy(z["da"].astype(np.int64)).to(dev)
        e1 = torch.from_numpy(z["ebwt"]).to(dev)
        copy = torch.arange(k, device=dev, dtype=torch.int64).repeat_interleave(m)
        dd = d1.repeat(k)
        isr = dd < nr1                                  # copy c: reads c*nr1 .., genomes (after ALL reads) c*ng1 ..
        rid = dd + copy * nr1
        if wl.get("spread"):
            rid = (rid * int(wl["spread"])) % (k * nr1)
        da.copy_(torch.where(isr, rid, k * nr1 + copy * ng1 + (dd - nr1)).to(torch.int32))
        lcp.copy_(l1.repeat(k)); eb.copy_(e1.repeat(k))
        del l1, d1, e1, copy, dd, isr, rid
    else:
        ctx.synth_dev(SEED, lo, n_avail, wl["nr"], wl["ng"], ALPHA, wl["mode"], lcp, da, eb, stream)
    torch.cuda.synchronize()
    counter = [0]
    ex_events = []                             # (start, end) around every exchange of the timed steps, on the stream it runs on

    def timed_exchange(fn, on_stream=None):
        if not ex_timing[0]:
            return fn()
        st_ = on_stream if on_stream is not None else torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st_); r_ = fn(); e1.record(st_)
        ex_events.append((e0, e1))
        return r_
    ex_timing = [False]

    def step():
        b = counter[0] % nbuf
        counter[0] += 1
        if done[b] is not None:                # the table buffer is free again (stream-side wait)
            torch.cuda.current_stream().wait_event(done[b])
            done[b] = None
        if sparse:
            # the updates leave the scan as records grouped by table bin; their owners build the table (lime_comm_exchange_records)
            ctx.fused_records_dev(lcp, da, eb, n_own, n_avail, hi_halo == n_total, wl["nr"], wl["ng"], ALPHA, stream)
            _s, _rc = ctx.stats(stream)
            if _rc:
                sys.exit(f"scan failed: rc={_rc}")
            timed_exchange(lambda: comm.exchange_records(ctx, wl["nr"], wl["ng"], own_block, stream))
            return
        ctx.fused_dev(lcp, da, eb, n_own, n_avail, hi_halo == n_total, wl["nr"], wl["ng"], ALPHA, sims[b], True, stream)
        if world > 1:
            # the pass is final only once lime_get_stats has returned (a record pool that proved too small is repaired there):
            # settle it before its table goes into the exchange
            _s, _rc = ctx.stats(stream)
            if _rc:
                sys.exit(f"scan failed: rc={_rc}")
            if ex_stream is None:              # exposed: the exchange follows the pass on the same stream
                timed_exchange(lambda: comm.reduce_scatter_tables(sims[b], blks[b], blk_bytes, stream))
            else:                              # overlapped: on its own stream, under the next step's scan
                ev = torch.cuda.Event(); ev.record()
                ex_stream.wait_event(ev)
                timed_exchange(lambda: comm.reduce_scatter_tables(sims[b], blks[b], blk_bytes, ex_stream.cuda_stream), ex_stream)
                done[b] = torch.cuda.Event(); done[b].record(ex_stream)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            comm.barrier()
        torch.cuda.synchronize()

    # ---- the COLD pass: the first one on a fresh context, as LiME_paired.sh:62-68 runs every collection -- once.  It pays the sampled
    # density probe, the scratch / record-pool allocations, and (if the probe misled it) a repeated pass; wall clock around pass + statistics
    cold = None
    if world == 1:
        torch.cuda.synchronize()
        tc0 = time.perf_counter()
        step()
        s, rc = ctx.stats(stream)
        torch.cuda.synchronize()
        cold_ms = (time.perf_counter() - tc0) * 1e3
        if rc:
            sys.exit(f"scan failed: rc={rc}")
        ht = ctx.host_times()
        cold = {"cold_ms": cold_ms, "alloc_ms": ht["alloc_ms"], "probe_ms": ht["probe_ms"], "probes": ht["probes"], "repeated_passes": ht["repeats"],
                "cas_fallbacks": ht["cas_fallbacks"], "records_per_symbol_probed": ht["records_per_symbol"],
                "update_path": "binned" if s.wave_records_max > 0 else "compare-and-swap", "flags": int(s.flags)}
    for _ in range(warmup):
        step()
        s, rc = ctx.stats(stream)              # settles the pass: update path, pool size (a pass that overflowed its record pool is repeated here)
        if rc:
            sys.exit(f"scan failed: rc={rc}")
    for _ in range(2 if warmup else 0):
        step()
    s, rc = ctx.stats(stream)
    ctx.set_timing(True)
    ex_timing[0] = world > 1
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    ex_timing[0] = False
    exchange_ms = sum(a.elapsed_time(b) for a, b in ex_events) / len(ex_events) if ex_events else None
    parts, launches = ctx.get_timing_ex()
    # the same passes once more, one at a time (HIP events around each): median and minimum next to the mean of the timed region
    each = []
    if world == 1:
        for _ in range(max(10, steps) if n_total <= 2_000_000_000 else max(5, steps)):
            step()
            p1, _n = ctx.get_timing_ex()
            each.append(p1["pass"])
    ctx.set_timing(False)
    s, rc = ctx.stats(stream)
    if rc:
        sys.exit(f"scan failed: rc={rc}")
    if cold is not None:
        ht = ctx.host_times()
        cold["repeated_passes_after_cold"] = ht["repeats"] - cold["repeated_passes"]
    each.sort()
    res = {"exchange": exchange, "dt": dt, "parts": parts, "exchange_ms": exchange_ms, "cold": cold, "flags": int(s.flags),
           "pass_ms_each": {"n": len(each), "median": each[len(each) // 2], "min": each[0], "max": each[-1]} if each else None, "launches": launches, "n_own": n_own, "n_clusters": int(s.n_clusters), "max_len": int(s.max_len),
           "updates": int(s.n_updates), "binned": bool(s.wave_records_max > 0), "wave_records_max": int(s.wave_records_max), "lcp": lcp, "da": da, "eb": eb}
    ctx.close()
    return res


def choose_flow(torch, lime_amd, dev, wname="c3"):
    """The device side of the reference's whole ClusterBWT_DA: the scoring pass + clusterChoose (ClusterBWT_DA.cpp:360-452: row maxima, the float
    test against beta, the surviving (idRef, sim) pairs on the host) through lime_fused_choose_dev, with the table in HBM (pass -> k_choose ->
    k_gather_pairs) and without it (the second-level kernel keeps row max / nnz and gathers the pairs from LDS; option choose_free).
    Wall clock of the call, results on the host; min and median of three calls each, alternating."""
    import time
    import numpy as np
    wl = WORKLOADS[wname]
    n, nr, ng = wl["n"], wl["nr"], wl["ng"]
    ctx = lime_amd.Context()
    lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
    eb = torch.empty(n, dtype=torch.uint8, device=dev) if wl["ebwt"] else None
    ctx.synth_dev(SEED, 0, n, nr, ng, ALPHA, wl["mode"], lcp, da, eb)
    res = {"workload": f"{wl['what']} through lime_fused_choose_dev (norm 85): scan + clusterAnalyze + clusterChoose, row maxima and pairs on the host; "
                       "beta 0.25: no row of the iid generator passes (the finish is the row maxima); beta 0.02: every row with a cell >= 2 passes, its pairs cross PCIe",
           "unit": "ms per call (wall clock, host results included)"}
    outs = (np.zeros(nr + 1, dtype=np.uint8), np.zeros(nr + 2, dtype=np.uint64))     # the caller's result arrays, as a C caller of the ABI holds them
    try:
        for beta in (0.25, 0.02):
            ms = {"0": [], "1": []}
            e = res["beta_%g" % beta] = {}
            for free in ("0", "1") * 4:                  # (the first call of each kind sizes its buffers: dropped)
                ctx.set_option("choose_free", free)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                rmx, roff, prs, st = ctx.fused_choose_dev(lcp, da, eb, n, nr, ng, ALPHA, 85, beta, out=outs)
                ms[free].append((time.perf_counter() - t0) * 1e3)
                e["pairs"] = int(len(prs)); res["n_clusters"] = int(st.n_clusters)
                del rmx, roff, prs
            for free, name in (("0", "with_table"), ("1", "without_table")):
                v = sorted(ms[free][1:])
                e[name] = {"min": round(v[0], 3), "median": round(v[len(v) // 2], 3), "symbols_per_s": n / (v[len(v) // 2] * 1e-3)}
            e["speedup_median"] = e["with_table"]["median"] / e["without_table"]["median"]
    finally:
        ctx.close(); del lcp, da, eb
    return res


def summarize(wl, r, n_total, steps):
    bps = 8 + wl["ebwt"]
    scan_ms, pass_ms = r["parts"]["scan"], r["parts"]["pass"]
    cold = dict(r["cold"]) if r.get("cold") else None
    if cold:
        steady = r["dt"] / steps * 1e3
        cold["over_steady"] = cold["cold_ms"] / steady if steady else None      # WITH the allocations (alloc_ms is a part of cold_ms, not taken out)
    return {"workload": describe(wl, n_total, 1), "ms_per_step": r["dt"] / steps * 1e3, "symbols_per_s": n_total * steps / r["dt"],
            "pass_ms_each": r.get("pass_ms_each"), "cold": cold,
            "pass_frac_median": bps * r["n_own"] / r["pass_ms_each"]["median"] / 1e6 / HBM_PEAK_GBS if r.get("pass_ms_each") else None,
            "update_path": "binned (records -> bins -> table regions built in LDS)" if r["binned"] else "compare-and-swap on the table",
            "kernel_ms_avg": scan_ms, "kernel_GBps": bps * r["n_own"] / scan_ms / 1e6 if scan_ms else None,
            "frac_of_hbm_peak": bps * r["n_own"] / scan_ms / 1e6 / HBM_PEAK_GBS if scan_ms else None,
            "pass_ms_avg": pass_ms, "pass_GBps": bps * r["n_own"] / pass_ms / 1e6 if pass_ms else None,
            "pass_frac_of_hbm_peak": bps * r["n_own"] / pass_ms / 1e6 / HBM_PEAK_GBS if pass_ms else None,
            "parts_ms": r["parts"], "n_clusters": r["n_clusters"], "table_updates": r["updates"],
            "records_fullest_wave_sub_region": r.get("wave_records_max")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS), help="default: c3 at N=1, n1e10 (strong) at N>1")
    ap.add_argument("--scaling", default=None, choices=["strong", "weak"], help="N>1: strong (fixed --n-total, default) or weak (workload per GPU)")
    ap.add_argument("--n-total", type=float, default=None, help="symbols of the whole collection (strong scaling)")
    ap.add_argument("--n", type=float, default=None, help="symbols per GPU (overrides the workload's size)")
    ap.add_argument("--exchange", default="dense", choices=["dense", "sparse", "auto"],
                    help="N>1: dense = a uint8 reduce-scatter of whole tables (default); sparse = owner-partitioned exchange of update records; "
                         "auto = whichever moves fewer bytes for this workload (decided from a probe pass)")
    ap.add_argument("--no-probe", action="store_true", help="profiling runs only (tools/profile.sh): no sampled density probe in front of the first pass -- it is a "
                    "short launch of the same k_scan instance and would pull the kernel's average in a trace 5 %% below that of a pass")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--cpu-sample", type=int, default=200_000_000)
    args = ap.parse_args()

    # The library reads its tuning / test knobs from the environment only under LIME_TEST_HOOKS=1 (lime_init): with that set -- or with another
    # build of the library (LIME_LIB), or ANY LIME_* variable this script does not know -- what is measured is not the release path
    allowed = {"LIME_BENCH_BACKEND", "LIME_IO_THREADS"}
    bad = sorted(k for k in os.environ if k.startswith("LIME_") and k not in allowed)
    if bad:
        sys.exit(f"bench.py refuses to run with {', '.join(bad)} set: it could change what is measured")

    import torch
    import lime_amd
    from lime_amd import dist as ldist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: there is no CPU path")
    # rehearsal hook for a one-GPU box (not used by the driver): LIME_BENCH_BACKEND=gloo runs the N>1 control flow
    # with every rank on GPU 0 and the exchange staged through the host (RCCL wants one GPU per rank)
    backend = os.environ.get("LIME_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    comm, comm_info = None, None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
            comm = ldist.Comm(rank, world, dev)             # RCCL through the C ABI (lime_comm_*); bootstrap over torch.distributed
        else:
            dist.init_process_group(backend)
            comm = ldist.HostComm(rank, world, dev)
        comm.check_uint8_sum_wraps()           # raises if RCCL's uint8 sum does not wrap modulo 256
        comm_info = {"backend": backend, "world_size": world, "nccl_comm_count": comm.count(), "uint8_sum_wraps": True}

    scaling = args.scaling or "strong"         # the series is strong scaling (fixed 10^10 symbols); on one GPU the label says which series the line belongs to
    wname = args.workload or ("c3" if (world == 1 or scaling == "weak") else "n1e10")
    wl = dict(WORKLOADS[wname])
    if world > 1 and scaling == "strong":
        n_total = int(args.n_total or wl["n"])
    else:
        n_total = int(args.n or wl["n"]) * world
    # The library's large buffers (record pool, binned records, scratch) for everything this run will do are taken from the driver ONCE, here
    # (lime_reserve): a hipMalloc that is served from pages some process has freed waits while the driver clears them -- 30 ms per GB, 0.4 .. 5 s for
    # the pools of a 10^10-symbol pass, and whether a box has such pages is its history, not this program's ("allocations").
    # Rounds 1-5 paid that inside the first pass of a workload (and subtracted it); a process that knows what it will run pays it at start-up.
    # What is reserved: 16 bytes per expected update record (margins included) + 0.3 bytes per symbol of the largest workload of the run.
    def lib_need(w, n_sym):
        return int(n_sym * ((0.25 if w["mode"] != 0 else 0.13) * 16 + 0.3)) + (2 << 30)
    planned = [(wl, n_total // world)]
    if world == 1 and not args.no_also:
        planned += [(WORKLOADS[k_], WORKLOADS[k_]["n"]) for k_ in ALSO_ORDER if k_ != wname]
    reserve_bytes = max(lib_need(w_, n_) for w_, n_ in planned)
    if world == 1 and not args.no_also:
        reserve_bytes = max(reserve_bytes, lib_need(WORKLOADS["c4_shape"], WORKLOADS["c4_shape"]["n"]) + WORKLOADS["c4_shape"]["nr"] * WORKLOADS["c4_shape"]["ng"])
    t_res = time.perf_counter()
    lime_amd.reserve(reserve_bytes)
    torch.cuda.synchronize()
    reserve_info = {"bytes": reserve_bytes, "ms": (time.perf_counter() - t_res) * 1e3,
                    "what": "lime_reserve at start-up: one block for the library's record pools / binned records / scratch of every workload of the run; "
                            "`cold` passes below carve from it (cold.alloc_ms is what they still spent in the allocator)"}
    r = run_pass_series(torch, lime_amd, ldist, wl, n_total, args.steps, args.warmup, world, rank, dev, comm, overlap=False, exchange=args.exchange,
                        options={"no_probe": "1"} if args.no_probe else None)
    dt, n_clusters, max_len = r["dt"], r["n_clusters"], r["max_len"]
    if world > 1:
        dt = comm.max_float(dt)
        n_clusters, max_len = comm.combine_counters(n_clusters, max_len)

    slow_parts = {k_: comm.max_float(float(v_)) for k_, v_ in sorted(r["parts"].items())} if world > 1 else None
    slow_ex = comm.max_float(float(r.get("exchange_ms") or 0.0)) if world > 1 else None
    out = None
    if rank == 0:
        bps = 8 + wl["ebwt"]
        scan_ms, pass_ms = r["parts"]["scan"], r["parts"]["pass"]
        k_achieved = bps * r["n_own"] / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
        step_ms = dt / args.steps * 1e3
        achieved = bps * r["n_own"] / (step_ms * 1e-3) / 1e9       # the WHOLE step (every kernel of a pass, N > 1: + the exchange): what `value` implies per GPU
        traffic, tsrc, pass_traffic = None, None, None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")        # rocprofv3 --pmc results, see DESIGN.md
        if os.path.exists(tfile) and world == 1:
            try:
                t = json.load(open(tfile)).get(wname)
                if t and t.get("symbols") == n_total:
                    traffic, tsrc = t["hbm_bytes_per_launch"], f"profiles/traffic.json[{wname}] ({t.get('from', '')}); file, not measured in this run"
                    pass_traffic = t.get("pass_hbm_bytes")
            except Exception:
                traffic = None
        # which instance of the scan ran: <E, 0, 0> compare-and-swap; binned: <E, 0, 2> where the scorers write finished records (tables of one or two
        # 4 GB sub-regions), <E, 0, 1> through the update queue (larger tables)
        kname = f"lime::k_scan<{wl['ebwt']}, 0, {(2 if lime_amd.sim_bytes(wl['nr'], wl['ng']) <= (2 << 32) else 1) if r['binned'] else 0}>"
        out = {
            "metric": "eBWT symbols/s processed (ClusterLCP+ClusterBWT_DA)", "value": n_total * args.steps / dt, "unit": "symbols/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": describe(wl, n_total, world), "symbols_total": n_total,
                       "sharding": ((f"position ranges x{world}; tables combined by one uint8 reduce-scatter per step through the C ABI "
                                     f"(RCCL), exposed (the step waits for it)") if r["exchange"] == "dense" else
                                    (f"position ranges x{world}; update records exchanged owner-partitioned through the C ABI (RCCL all-gather + "
                                     f"send/receive), every rank builds its block of the table")) if world > 1 else "one GPU",
                       "update_path": "binned (records -> bins -> table regions built in LDS)" if r["binned"] else "compare-and-swap on the table",
                       "n_clusters": int(n_clusters), "max_cluster_len": int(max_len), "table_updates_rank0": r["updates"]},
            # achieved / frac describe the STEP (= what `value` implies: algorithmic bytes of a pass / ms_per_step); the dominant kernel alone
            # (k_scan, HIP events on its stream) is under kernel_*.  traffic = HBM bytes of ALL kernels of a pass, kernel_traffic of k_scan alone.
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "frac_of": "the whole step (all kernels of a pass; N > 1: + the exchange) -- value x bytes per symbol / peak per GPU",
                         "traffic": pass_traffic, "traffic_source": tsrc, "algorithmic_bytes_per_launch": bps * r["n_own"],
                         "kernel": kname, "kernel_ms_avg": scan_ms, "kernel_achieved": k_achieved, "kernel_frac": k_achieved / HBM_PEAK_GBS,
                         "kernel_traffic": traffic, "launches_timed": r["launches"],
                         "pass_ms_avg": pass_ms, "pass_achieved": bps * r["n_own"] / (pass_ms * 1e-3) / 1e9 if pass_ms else None,
                         "pass_frac": bps * r["n_own"] / (pass_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if pass_ms else None,
                         "pass_ms_each": r.get("pass_ms_each"), "cold": r.get("cold"),
                         "pass_parts_ms": r["parts"], "pass_parts_ms_slowest_rank": slow_parts, "exchange_ms_slowest_rank": slow_ex},
        }
    if world == 1 and rank == 0 and not args.no_cpu:
        try:
            out["cpu_baseline"] = cpu_baseline(wl, r["lcp"], r["da"], r["eb"], r["n_own"], args.cpu_sample)
        except Exception as e:   # a missing baseline must not hide the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "symbols/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
    first_exchange = r["exchange"]
    headline_parts, headline_ex_ms = r["parts"], r.get("exchange_ms")
    del r

    def make_room(w, extra=0):
        # torch keeps the blocks of the workload before (and reuses them where they fit); the library's blocks come out of the block reserved at start-up.
        # Nothing goes back to the driver between workloads unless the next one would not fit: memory that is freed comes back uncleared, and the next
        # hipMalloc that gets it waits for the driver to clear it (tools/alloc_bench.hip) -- freeing 90 GB of arrays in front of
        # every workload, as rounds 1-5 did, made the library's first allocations of the next one take between 1 ms and 5 s (cold.alloc_ms).
        # (round 6, later: the library's blocks come out of the block reserved at start-up; what must be free at the driver is only a margin for
        # torch's own allocations of the next workload beside its cached blocks -- it releases its cache by itself when one of them does not fit)
        free_b, _tot = torch.cuda.mem_get_info()
        if free_b < (4 << 30):
            torch.cuda.empty_cache()

    if not args.no_also:
        also = {}
        if world == 1:
            # (the workloads with the largest arrays first: torch's cached blocks are reused by everything after them)
            for name in ALSO_ORDER:
                if name == wname:
                    continue
                w2 = WORKLOADS[name]
                try:
                    make_room(w2)
                    k = max(5, args.steps // 2) if w2["n"] >= 10_000_000_000 else 3 if name == "c4_shape" else max(10, args.steps)
                    r2 = run_pass_series(torch, lime_amd, ldist, w2, w2["n"], k, 2, 1, 0, dev, None, overlap=False)
                    also[name] = summarize(w2, r2, w2["n"], k)
                    del r2
                except Exception as e:
                    also[name] = {"failed": str(e)}
            # scan + clusterAnalyze + clusterChoose with and without the table: configs[2], and the shape of configs[3] (setB2's 18.8 GB table,
            # ClusterBWT_DA.cpp:385-443 would scan it single-threaded)
            for name, key in (("c3", "c3_with_choose"), ("c4_shape", "c4_with_choose")):
                try:
                    make_room(WORKLOADS[name], WORKLOADS[name]["nr"] * WORKLOADS[name]["ng"])      # (the with-table finish allocates the table in the library)
                    also[key] = choose_flow(torch, lime_amd, dev, name)
                except Exception as e:
                    also[key] = {"failed": str(e)}
        else:
            # N > 1, in this ONE invocation (the driver's scaling run is the only multi-GPU run there is): the dense reduce-scatter exposed (the
            # headline above), the same overlapped, the owner-partitioned exchange of update records, and whichever of the two a probe pass picks
            def series(overlap, exchange, what):
                r_ = run_pass_series(torch, lime_amd, ldist, wl, n_total, args.steps, args.warmup, world, rank, dev, comm, overlap=overlap, exchange=exchange)
                dt_ = comm.max_float(r_["dt"])
                e_ = {"what": what, "exchange": r_["exchange"], "value": n_total * args.steps / dt_, "ms_per_step": dt_ / args.steps * 1e3,
                      "parts_ms_slowest_rank": {k_: comm.max_float(float(v_)) for k_, v_ in sorted(r_["parts"].items())},
                      "exchange_ms_slowest_rank": comm.max_float(float(r_["exchange_ms"] or 0.0)), "table_updates_rank0": r_["updates"]}
                del r_                                  # (its blocks stay with torch: the next series asks for the same sizes)
                return e_
            for key, overlap, exch, what in (
                    ("overlapped", True, "dense", "the dense series with the exchange of step k under the scan of step k+1 (two table buffers)"),
                    ("sparse_exchange", False, "sparse", "the series with the owner-partitioned exchange of update records instead of the dense reduce-scatter"),
                    ("auto", False, "auto", "the series with the exchange a probe pass picks (dist.choose_exchange: 4 bytes x the largest update count of a rank against the table's bytes)")):
                if exch == first_exchange and not overlap:
                    continue
                try:
                    also[key] = series(overlap, exch, what)
                except Exception as e:
                    also[key] = {"failed": str(e)}
        if rank == 0:
            out["also"] = also
    if rank == 0 and world > 1:
        out["comm"] = comm_info
        out["roofline"]["exchange_ms_rank0"] = headline_ex_ms
    if rank == 0:
        out["reserve"] = reserve_info
        print(json.dumps(out), flush=True)
    if world > 1:
        comm.close()
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
