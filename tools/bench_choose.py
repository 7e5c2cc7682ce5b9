#!/usr/bin/env python3
"""tools/bench_choose.py [n_reads n_refs density] -- clusterChoose on a device-resident table:
k_choose (row max / nnz) + host pass test and prefix + k_gather_pairs + D2H of the compact lists,
against the bytes of the table.  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lime_amd

nr = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
ng = int(float(sys.argv[2])) if len(sys.argv) > 2 else 5000
dens = float(sys.argv[3]) if len(sys.argv) > 3 else 0.002
ctx = lime_amd.Context()
dev = torch.device("cuda:0")
T = nr * ng
buf = torch.zeros(lime_amd.sim_bytes(nr, ng), dtype=torch.uint8, device=dev)
g = torch.Generator(device=dev); g.manual_seed(1)
step = 1 << 28
for o in range(0, T, step):                       # sparse random table, built in slices
    m = min(step, T - o)
    v = torch.randint(1, 256, (m,), dtype=torch.int16, device=dev, generator=g).to(torch.uint8)
    keep = torch.rand(m, device=dev, generator=g) < dens
    buf[o:o + m] = v * keep
    del v, keep
torch.cuda.synchronize()
mx = torch.empty(nr, dtype=torch.uint8, device=dev); nz = torch.empty(nr, dtype=torch.int32, device=dev)
for _ in range(2):
    ctx.choose_dev(buf, nr, ng, mx, nz)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ctx.choose_dev(buf, nr, ng, mx, nz)
e1.record(); torch.cuda.synchronize()
row_ms = e0.elapsed_time(e1) / 5
t0 = time.perf_counter()
rmax, off, pairs = ctx.choose_pairs_dev(buf, nr, ng, 85, 0.25)
whole_s = time.perf_counter() - t0
# ---- the whole ClusterBWT_DA-side of a device-resident pass: scan + table + k_choose + k_gather_pairs against the same call without the table
# (lime_fused_choose_dev, LIME_CHOOSE_FREE=0 / 1), BASELINE.json configs[2]'s input, two values of beta
fc = {}
if nr == 1_000_000 and ng == 5000 and not os.environ.get("CHOOSE_ONLY"):
    del buf, mx, nz
    torch.cuda.empty_cache()
    n = 1_000_000_000
    lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
    ctx.synth_dev(42, 0, n, nr, ng, 16, 0, lcp, da, None)
    for beta in (0.25, 0.02):
        for free in ("0", "1", "0", "1"):
            os.environ["LIME_CHOOSE_FREE"] = free
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rmx, roff, prs, st = ctx.fused_choose_dev(lcp, da, None, n, nr, ng, 16, 85, beta)
            dt = time.perf_counter() - t0
            fc.setdefault(f"beta_{beta}", {}).setdefault("without_table" if free == "1" else "with_table", []).append(round(dt * 1e3, 3))
            fc[f"beta_{beta}"]["pairs"] = int(len(prs))
    del os.environ["LIME_CHOOSE_FREE"]
    for v in fc.values():
        v["speedup_min_over_min"] = min(v["with_table"]) / min(v["without_table"])
    buf = torch.zeros(16, dtype=torch.uint8, device=dev); pairs = prs
print(json.dumps({"fused_choose_ms": fc, "table": f"{nr}x{ng}", "table_bytes": T, "density": dens, "k_choose_ms": row_ms,
                  "k_choose_GBps": T / row_ms / 1e6, "pairs": int(len(pairs)),
                  "choose_pairs_total_s": whole_s, "choose_pairs_GBps_of_table": T / whole_s / 1e9}))
