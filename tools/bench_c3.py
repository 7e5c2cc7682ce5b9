#!/usr/bin/env python3
"""tools/bench_c3.py -- BASELINE.json configs[2] (10^9 symbols, 10^6 reads x 5000 genomes = 5 GB table,
EBWT=0, alpha=16) on one MI355X: one fused pass per update path (LIME_UPDATE_PATH=cas|bin), parts of the pass
from HIP events.  Quick A/B tool; the headline line is bench.py's."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lime_amd

n = int(os.environ.get("C3_N", 1_000_000_000))
nr, ng, alpha = int(os.environ.get("C3_NR", 1_000_000)), int(os.environ.get("C3_NG", 5000)), 16
mode = int(os.environ.get("C3_MODE", 0))
dev = torch.device("cuda:0")
lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
eb = torch.empty(n, dtype=torch.uint8, device=dev) if int(os.environ.get("C3_EBWT", 0)) else None   # C3_EBWT=1 C3_N=100000000 C3_NR=100000 C3_NG=500: configs[1]
bps = 9 if eb is not None else 8
res = {"symbols": n, "table_bytes": nr * ng, "mode": mode}
ref = None
for path in os.environ.get("C3_PATHS", "cas,bin").split(","):
    os.environ["LIME_UPDATE_PATH"] = path
    ctx = lime_amd.Context()
    if ref is None:
        ctx.synth_dev(42, 0, n, nr, ng, alpha, mode, lcp, da, eb)
    sim = torch.empty(lime_amd.sim_bytes(nr, ng), dtype=torch.uint8, device=dev)
    for _ in range(2):
        ctx.fused_dev(lcp, da, eb, n, n, True, nr, ng, alpha, sim, True)
        s, rc = ctx.stats(); assert rc == 0, rc
    ctx.set_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        ctx.fused_dev(lcp, da, eb, n, n, True, nr, ng, alpha, sim, True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    parts, launches = ctx.get_timing_ex(); ctx.set_timing(False)
    s, rc = ctx.stats(); assert rc == 0
    res[path] = {"ms_per_pass": dt * 1e3, "symbols_per_s": n / dt, "parts_ms": parts, "k_scan_GBps": bps * n / parts["scan"] / 1e6,
                 "pass_GBps": bps * n / parts["pass"] / 1e6, "wave_records_max": s.wave_records_max}
    res["n_clusters_" + path], res["table_updates_" + path] = int(s.n_clusters), int(s.n_updates)
    if ref is None:
        ref = sim.clone()
    else:
        res["tables_equal"] = bool(torch.equal(ref, sim))
    ctx.close(); del sim
print(json.dumps(res))
