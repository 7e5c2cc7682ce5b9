#!/usr/bin/env python3
"""tools/bench_c3.py -- BASELINE.json configs[2] (10^9 symbols, 10^6 reads x 5000 genomes = 5 GB table,
EBWT=0, alpha=16) on one MI355X: one fused pass with and without clearing the table.  Secondary number for
DESIGN.md; the headline metric stays bench.py (configs[1])."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lime_amd

n, nr, ng, alpha = 1_000_000_000, 1_000_000, 5000, 16
ctx = lime_amd.Context()
dev = torch.device("cuda:0")
lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
ctx.synth_dev(42, 0, n, nr, ng, alpha, 0, lcp, da, None)
sim = torch.empty(lime_amd.sim_bytes(nr, ng), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
res = {"symbols": n, "table_bytes": nr * ng}
for zero in (True, False):
    for _ in range(2):
        ctx.fused_dev(lcp, da, None, n, n, True, nr, ng, alpha, sim, zero)
    s, rc = ctx.stats(); assert rc == 0
    ctx.set_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        ctx.fused_dev(lcp, da, None, n, n, True, nr, ng, alpha, sim, zero)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    scan_ms, launches = ctx.get_timing(); ctx.set_timing(False)
    res["with_table_clear" if zero else "no_table_clear"] = {"ms_per_pass": dt * 1e3, "symbols_per_s": n / dt, "k_scan_ms": scan_ms,
                                                              "k_scan_GBps": 8 * n / scan_ms / 1e6}
res["n_clusters"], res["table_updates"] = int(s.n_clusters), int(s.n_updates)
print(json.dumps(res))
