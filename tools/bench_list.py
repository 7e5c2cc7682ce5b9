#!/usr/bin/env python3
"""tools/bench_list.py -- the two-program flow with the arrays resident (lime_detect_dev, then lime_score_dev on the cluster list) on the
clustered generator: list scoring by compare-and-swap (LIME_UPDATE_PATH=cas) against the binned path (bin).  ms per call, tables compared."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lime_amd

n = int(os.environ.get("L_N", 100_000_000)); nr = int(os.environ.get("L_NR", 452_000)); ng = int(os.environ.get("L_NG", 678))
dev = torch.device("cuda:0")
lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp); eb = torch.empty(n, dtype=torch.uint8, device=dev)
ref = None
for path in ("cas", "bin"):
    os.environ["LIME_UPDATE_PATH"] = path
    ctx = lime_amd.Context()
    ctx.synth_dev(42, 0, n, nr, ng, 16, 1, lcp, da, eb)
    dc, nc, ml = ctx.detect_dev(lcp, da, n, n, True, 0, nr, 16)
    torch.cuda.synchronize()
    sim = torch.empty(lime_amd.sim_bytes(nr, ng), dtype=torch.uint8, device=dev)
    for _ in range(2):
        ctx.score_dev(da, eb, n, dc, nc, nr, ng, sim, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        ctx.score_dev(da, eb, n, dc, nc, nr, ng, sim, True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    s, rc = ctx.stats(); assert rc == 0, rc
    print(f"{path}: {dt * 1e3:.3f} ms per lime_score_dev ({n} symbols, {nc} clusters, {int(s.n_updates)} updates, table {nr}x{ng}; records per wave max {s.wave_records_max})")
    if ref is None: ref = sim.clone()
    else: print("tables equal:", bool(torch.equal(ref, sim)))
    ctx.close()
