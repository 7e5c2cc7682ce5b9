#!/usr/bin/env python3
"""tools/bench_cli.py -- end-to-end rate of the drop-in programs on files (page cache -> pinned ring -> HBM -> kernels ->
output files): ClusterLCP then ClusterBWT_DA on synthetic S of C_N symbols (default 4e8) written to /tmp.  PCIe-inclusive:
never bench.py's `value`."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lime_amd

n = int(float(os.environ.get("C_N", 4e8))); nr, ng, alpha = 1_000_000, 500, 16
res = {"symbols": n, "reads": nr, "genomes": ng, "what": "wall clock of the drop-in programs, process start included; phases from LIME_CLI_TIMING=1; files in the page cache"}
import shutil
need = 9 * n + n // 2 * 16 + (1 << 30)
if shutil.disk_usage("/tmp").free < need:
    sys.exit(f"bench_cli: /tmp has {shutil.disk_usage('/tmp').free >> 30} GB free, {need >> 30} GB needed for {n} symbols")
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    base = os.path.join(td, "S.fasta")
    ctx = lime_amd.Context(0)
    dev = torch.device("cuda:0")
    lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp); eb = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.synth_dev(42, 0, n, nr, ng, alpha, 0, lcp, da, eb); torch.cuda.synchronize()
    for t, ext in ((lcp, ".lcp"), (da, ".da"), (eb, ".ebwt")):
        with open(base + ext, "wb") as f:                 # in pieces: no second copy of 16 GB on the host
            for lo in range(0, n, 1 << 27):
                f.write(t[lo:min(n, lo + (1 << 27))].cpu().numpy().tobytes())
    del lcp, da, eb; ctx.close(); torch.cuda.empty_cache(); lime_amd.trim_cache()
    for threads in (8, 16, 32, 4):
        env = {k: v for k, v in os.environ.items() if not k.startswith("LIME_")}
        env["LIME_CLI_TIMING"] = "1"
        t0 = time.perf_counter()
        p1 = subprocess.run([f"{ROOT}/lime_amd/bin/ClusterLCP", base, str(nr), str(ng), str(alpha), str(threads)], check=True, capture_output=True, env=env, cwd=td)
        t1 = time.perf_counter()
        p2 = subprocess.run([f"{ROOT}/lime_amd/bin/ClusterBWT_DA", base, "100", "0.25", str(threads)], check=True, capture_output=True, env=env, cwd=td)
        t2 = time.perf_counter()
        marks = lambda p: [ln.strip() for ln in p.stderr.decode(errors="replace").splitlines() if ln.startswith("[cli]")]
        ph = {"ClusterLCP": marks(p1), "ClusterBWT_DA": marks(p2)}
        def ms_of(lines, key):
            for ln in lines:
                if key in ln:
                    return float(ln.split(key)[1].split("ms")[0])
            return None
        w1, w2 = ms_of(ph["ClusterLCP"], "scan + .clrs"), ms_of(ph["ClusterBWT_DA"], "scoring + choose")
        clrs_bytes = os.path.getsize(f"{base}.{alpha}.clrs")
        res[f"threads_{threads}"] = {"ClusterLCP_s": t1 - t0, "ClusterBWT_DA_s": t2 - t1, "symbols_per_s": n / (t2 - t0),
                                     # the working phases alone (what the link and the host's page-cache copies bound), against 53 GB/s from pinned memory
                                     "ClusterLCP_scan_phase_GBps_in": 8 * n / (w1 * 1e6) if w1 else None,
                                     "ClusterBWT_DA_scoring_phase_GBps_in": (5 * n + clrs_bytes) / (w2 * 1e6) if w2 else None,
                                     "clrs_bytes": clrs_bytes, "phases": ph}
print(json.dumps(res))
