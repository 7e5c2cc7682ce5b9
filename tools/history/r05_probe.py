#!/usr/bin/env python3
"""probe density vs counted density, repeats, on synthetic shapes (debug aid, round 5)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, lime_amd
dev = torch.device("cuda", 0)
shapes = [(int(float(a)), int(b), int(c), int(d), int(e)) for a, b, c, d, e in (x.split(",") for x in sys.argv[1:])] or \
    [(1e9, 1000000, 1000, 0, 1), (1e9, 3000000, 3423, 1, 1), (1e8, 100000, 500, 1, 0), (1e8, 100000, 500, 1, 1)]
for n, nr, ng, ebwt, mode in shapes:
    n = int(n)
    c = lime_amd.Context()
    lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
    eb = torch.empty(n, dtype=torch.uint8, device=dev) if ebwt else None
    c.synth_dev(42, 0, n, nr, ng, 16, mode, lcp, da, eb)
    sim = torch.empty(lime_amd.sim_bytes(nr, ng), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c.fused_dev(lcp, da, eb, n, n, True, nr, ng, 16, sim, True)
    h0 = c.host_times()
    s, rc = c.stats()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    h1 = c.host_times()
    print(f"n={n} {nr}x{ng} ebwt={ebwt} mode={mode}: rc={rc} probed={h0['records_per_symbol']} counted={s.n_updates / n:.5f} wave_records_max={s.wave_records_max} "
          f"repeats={h1['repeats']} fallbacks={h1['cas_fallbacks']} cold_ms={(t1 - t0) * 1e3:.2f} alloc_ms={h1['alloc_ms']:.2f} probe_ms={h1['probe_ms']:.3f} flags={s.flags}", flush=True)
    c.set_timing(True)
    for _ in range(3):
        c.fused_dev(lcp, da, eb, n, n, True, nr, ng, 16, sim, True)
    p, k = c.get_timing_ex()
    print("   steady:", {a: round(b, 3) for a, b in p.items()}, flush=True)
    c.close(); del lcp, da, eb, sim; torch.cuda.empty_cache()
