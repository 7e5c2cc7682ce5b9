#!/usr/bin/env python3
"""tools/bench_detect.py [n] -- detection only (the ClusterLCP program's work) on
device-resident arrays: count pass + resolve + prefix sum + ordered emission of (pStart, len) records.
Algorithmic bytes: 8 B/symbol read once + 16 B per record written."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lime_amd

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
nr, ng, alpha = 100000, 500, 16
ctx = lime_amd.Context()
dev = torch.device("cuda:0")
lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
ctx.synth_dev(42, 0, n, nr, ng, alpha, 0, lcp, da, None)
torch.cuda.synchronize()
for _ in range(3):
    ptr, nc, ml = ctx.detect_dev(lcp, da, n, n, True, 0, nr, alpha)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for _ in range(K):
    ptr, nc, ml = ctx.detect_dev(lcp, da, n, n, True, 0, nr, alpha)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(json.dumps({"symbols": n, "n_clusters": nc, "max_len": ml, "ms_per_pass": dt * 1e3, "symbols_per_s": n / dt,
                  "GBps_algorithmic": (8 * n + 16 * nc) / dt / 1e9}))
