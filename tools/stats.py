#!/usr/bin/env python3
"""tools/stats.py -- counters of one fused pass over the bench workload (device-generated)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lime_amd.api import Context, sim_bytes
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
nr, ng, alpha = 100000, 500, 16
ctx = Context(0)
dev = torch.device("cuda:0")
lcp = torch.empty(n + 64, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
eb = torch.empty(n + 64, dtype=torch.uint8, device=dev)
ctx.synth_dev(42, 0, n, nr, ng, alpha, mode, lcp, da, eb)
sim = torch.zeros(sim_bytes(nr, ng), dtype=torch.uint8, device=dev)
ctx.fused_dev(lcp, da, eb, n, n, 1, nr, ng, alpha, sim, True)
torch.cuda.synchronize()
st, rc = ctx.stats()
print({k: getattr(st, k) for k, _ in st._fields_})
