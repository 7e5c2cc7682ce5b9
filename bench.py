#!/usr/bin/env python3
"""bench.py -- throughput of the LiME hot path (ClusterLCP + ClusterBWT_DA) on MI355X.

Metric (BASELINE.json): eBWT symbols/s processed by detect+score, arrays resident in HBM.
Workload at N=1: BASELINE.json configs[1] -- synthetic S of 10^8 symbols, 10^5 reads x 500
genomes, alpha=16, EBWT=1 (generator: SURVEY.md 8d, seed 42, identical on CPU and GPU).
A step = zero the score table + one fused scan of the rank's shard (+ for N>1 the one
exchange of the path: a reduce-scatter of the per-rank uint8 tables, sum mod 256, by read-row
blocks -- SURVEY 8e -- issued asynchronously so that it runs under the next step's scan; two
table buffers alternate).
N>1 (launched by torch.distributed.run, one rank per GPU): WEAK scaling -- every rank owns
10^8 symbols of a 10^8*N collection cut by contiguous tile-aligned position ranges with a
read-ahead halo; no other data-path collective.

Prints ONE JSON line on rank 0 (see the driver's contract) with two extra objects:
  roofline     HBM bound: algorithmic bytes (9 B/symbol x symbols per launch) / average
               duration of the scan kernel k_tile measured with HIP events on its stream
  cpu_baseline the reference's own OpenMP programs (oracle/_ref, kind "reference") or the
               oracle port, timed on this box's host cores on the same workload
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PER_GPU = 100_000_000
N_READS, N_REFS, ALPHA, SEED = 100_000, 500, 16, 42
BYTES_PER_SYMBOL = 9            # lcp 4 + da 4 + ebwt 1 (EBWT=1), each input byte read once
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(lcp_t, da_t, eb_t, n, sample_n):
    """Reference binaries (or the oracle port) on the first `sample_n` symbols, all host cores."""
    import numpy as np
    # the GPU box gives one GPU's share of the host: at most 16 cores (the pool's sizing rule)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    sample_n = min(sample_n, n)
    lcp = lcp_t[:sample_n].cpu().numpy().view(np.uint32)
    da = da_t[:sample_n].cpu().numpy().view(np.uint32)
    eb = eb_t[:sample_n].cpu().numpy()
    ref = os.path.join(ROOT, "oracle", "_ref")
    sample = f"first {sample_n} symbols of the bench workload (seed {SEED}), {N_READS}x{N_REFS}, alpha {ALPHA}, EBWT=1"
    if os.path.exists(os.path.join(ref, "ClusterLCP")) and os.path.exists(os.path.join(ref, "ClusterBWT_DA")):
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            base = os.path.join(td, "S.fasta")
            lcp.tofile(base + ".lcp"); da.tofile(base + ".da"); eb.tofile(base + ".ebwt")
            t0 = time.perf_counter()
            subprocess.run([f"{ref}/ClusterLCP", base, str(N_READS), str(N_REFS), str(ALPHA), str(cores)],
                           check=True, capture_output=True, cwd=td, timeout=900)
            t1 = time.perf_counter()
            subprocess.run([f"{ref}/ClusterBWT_DA", base, "100", "0.25", str(cores)],
                           check=True, capture_output=True, cwd=td, timeout=900)
            t2 = time.perf_counter()
        return {"value": sample_n / (t2 - t0), "unit": "symbols/s", "cores": cores, "kind": "reference",
                "sample": sample + "; wall time of the reference's ClusterLCP + ClusterBWT_DA processes, files in page cache",
                "seconds": {"ClusterLCP": t1 - t0, "ClusterBWT_DA": t2 - t1}}
    from oracle import oracle_py as O
    t0 = time.perf_counter()
    cl, nc, ml = O.detect(lcp, da, N_READS, ALPHA)
    O.score(da, eb, cl, N_READS, N_REFS, threads=cores)
    t1 = time.perf_counter()
    return {"value": sample_n / (t1 - t0), "unit": "symbols/s", "cores": cores, "kind": "port", "sample": sample}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=N_PER_GPU, help="symbols per GPU")
    ap.add_argument("--mode", type=int, default=0, help="synthetic generator: 0 iid (configs[1]), 1 block-correlated")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-sample", type=int, default=N_PER_GPU)
    args = ap.parse_args()

    import torch
    import lime_amd
    from lime_amd import dist as ldist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: there is no CPU path")
    # rehearsal hooks for a one-GPU box (not used by the driver): LIME_BENCH_BACKEND=gloo runs the N>1 control
    # flow with every rank on GPU 0 and the exchange staged through the host
    backend = os.environ.get("LIME_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_rs = False
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        use_rs = ldist.check_uint8_sum_wraps(dev)

    n_total = args.n * world
    lo, hi, hi_halo = ldist.shard_ranges(n_total, world)[rank]
    n_own, n_avail = hi - lo, hi_halo - lo
    ctx = lime_amd.Context(local)
    lcp = torch.empty(n_avail, dtype=torch.int32, device=dev)
    da = torch.empty(n_avail, dtype=torch.int32, device=dev)
    eb = torch.empty(n_avail, dtype=torch.uint8, device=dev)
    sim_bytes = lime_amd.sim_bytes(N_READS, N_REFS)
    blk_bytes = ldist.table_block_bytes(sim_bytes, world)
    nbuf = 2 if world > 1 else 1
    sims = [torch.zeros(blk_bytes * world, dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    blks = [torch.empty(blk_bytes, dtype=torch.uint8, device=dev) for _ in range(nbuf)] if world > 1 else []
    pending = [None] * nbuf
    sim = sims[0]
    stream = torch.cuda.current_stream().cuda_stream
    ctx.synth_dev(SEED, lo, n_avail, N_READS, N_REFS, ALPHA, args.mode, lcp, da, eb, stream)
    torch.cuda.synchronize()

    counter = [0]

    def step():
        b = counter[0] % nbuf
        counter[0] += 1
        if pending[b] is not None:            # the exchange that read this buffer two steps ago (stream-side wait)
            pending[b].wait()
            pending[b] = None
        ctx.fused_dev(lcp, da, eb, n_own, n_avail, hi_halo == n_total, N_READS, N_REFS, ALPHA, sims[b], True, stream)
        if world > 1:
            if use_rs:
                pending[b] = ldist.reduce_scatter_tables(sims[b], blks[b], async_op=True)
            else:                             # backend without a wrapping uint8 reduce-scatter: whole-table all-reduce
                pending[b] = dist.all_reduce(sims[b], op=dist.ReduceOp.SUM, async_op=True)

    def barrier():
        for b in range(nbuf):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    s, rc = ctx.stats(stream)
    if rc:
        sys.exit(f"scan failed: rc={rc}")
    ctx.set_timing(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    scan_ms, launches = ctx.get_timing()
    ctx.set_timing(False)
    s, rc = ctx.stats(stream)
    n_clusters, max_len = s.n_clusters, s.max_len
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        n_clusters, max_len = ldist.combine_counters(n_clusters, max_len, dev)

    if rank == 0:
        value = n_total * args.steps / dt
        achieved = BYTES_PER_SYMBOL * n_own / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")     # rocprofv3 --pmc result, see DESIGN.md
        if os.path.exists(tfile) and args.n == N_PER_GPU and args.mode == 0:
            try:
                traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "eBWT symbols/s processed (ClusterLCP+ClusterBWT_DA)", "value": value, "unit": "symbols/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"synthetic S (seed {SEED}, generator mode {args.mode}): {args.n} symbols per GPU, "
                                   f"{N_READS} reads x {N_REFS} genomes, alpha={ALPHA}, EBWT=1 (BASELINE.json configs[1])",
                       "symbols_total": n_total, "sharding": f"position ranges x{world}" + ((f"; tables combined by an asynchronous uint8 {'reduce-scatter' if use_rs else 'all-reduce'} per step") if world > 1 else ""), "n_clusters": int(n_clusters),
                       "max_cluster_len": int(max_len), "table_updates": int(s.n_updates)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": "lime::k_scan<1, 0>",
                         "kernel_ms_avg": scan_ms, "launches_timed": launches,
                         "algorithmic_bytes_per_launch": BYTES_PER_SYMBOL * n_own},
        }
        if world == 1 and not args.no_cpu:
            try:
                out["cpu_baseline"] = cpu_baseline(lcp, da, eb, n_own, args.cpu_sample)
            except Exception as e:   # a missing baseline must not hide the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "symbols/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
