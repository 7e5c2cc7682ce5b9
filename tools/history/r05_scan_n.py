#!/usr/bin/env python3
"""scan rate against the input's length around 1e9 symbols (is the N-dependence of the scan's HBM fraction an address effect?)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, lime_amd
dev = torch.device("cuda", 0)
ns = [int(float(x)) for x in sys.argv[1:]] or [1000000000, 1000000000 + 4096 * 37, 1000000000 + 1024 * 1024 * 3 + 4096, 950000000, 1073741824, 1073741824 + 65536 * 5, 1100000000, 1500000000]
for n in ns:
    c = lime_amd.Context()
    lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
    c.synth_dev(42, 0, n, 1000000, 5000, 16, 0, lcp, da, None)
    sim = torch.empty(lime_amd.sim_bytes(1000000, 5000), dtype=torch.uint8, device=dev)
    for _ in range(2): c.fused_dev(lcp, da, None, n, n, True, 1000000, 5000, 16, sim, True); c.stats()
    c.set_timing(True)
    for _ in range(5): c.fused_dev(lcp, da, None, n, n, True, 1000000, 5000, 16, sim, True)
    p, k = c.get_timing_ex()
    print(f"n={n} lcp@{lcp.data_ptr():#x} da@{da.data_ptr():#x} (da-lcp) mod 2^20 = {(da.data_ptr() - lcp.data_ptr()) % (1 << 20)}: scan {p['scan']:.3f} ms = {8 * n / p['scan'] / 1e9:.2f} TB/s, pass {p['pass']:.3f}", flush=True)
    c.close(); del lcp, da, sim; torch.cuda.empty_cache()
