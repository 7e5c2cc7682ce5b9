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
res = {"symbols": n, "reads": nr, "genomes": ng}
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    base = os.path.join(td, "S.fasta")
    ctx = lime_amd.Context(0)
    dev = torch.device("cuda:0")
    lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp); eb = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.synth_dev(42, 0, n, nr, ng, alpha, 0, lcp, da, eb); torch.cuda.synchronize()
    lcp.cpu().numpy().tofile(base + ".lcp"); da.cpu().numpy().tofile(base + ".da"); eb.cpu().numpy().tofile(base + ".ebwt")
    del lcp, da, eb; ctx.close(); torch.cuda.empty_cache()
    for threads, staging in ((8, "ring"), (16, "ring"), (32, "ring"), (1, "ring"), (8, "direct")):
            env = dict(os.environ)
            env["LIME_CLI_TIMING"] = "1"
            if staging == "direct":
                env["LIME_NO_STAGING"] = "1"
            t0 = time.perf_counter()
            p1 = subprocess.run([f"{ROOT}/lime_amd/bin/ClusterLCP", base, str(nr), str(ng), str(alpha), str(threads)], check=True, capture_output=True, env=env, cwd=td)
            t1 = time.perf_counter()
            p2 = subprocess.run([f"{ROOT}/lime_amd/bin/ClusterBWT_DA", base, "100", "0.25", str(threads)], check=True, capture_output=True, env=env, cwd=td)
            t2 = time.perf_counter()
            marks = lambda p: [ln.strip() for ln in p.stderr.decode(errors="replace").splitlines() if ln.startswith("[cli]")]
            res[f"{staging}_t{threads}"] = {"ClusterLCP_s": t1 - t0, "ClusterLCP_GBps": 8 * n / (t1 - t0) / 1e9,
                                            "ClusterBWT_DA_s": t2 - t1, "ClusterBWT_DA_GBps": 5 * n / (t2 - t1) / 1e9,
                                            "symbols_per_s": n / (t2 - t0), "phases": {"ClusterLCP": marks(p1), "ClusterBWT_DA": marks(p2)}}
print(json.dumps(res))
