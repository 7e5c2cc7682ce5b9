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
print(json.dumps({"table": f"{nr}x{ng}", "table_bytes": T, "density": dens, "k_choose_ms": row_ms,
                  "k_choose_GBps": T / row_ms / 1e6, "pairs": int(len(pairs)),
                  "choose_pairs_total_s": whole_s, "choose_pairs_GBps_of_table": T / whole_s / 1e9}))
