#!/usr/bin/env python3
"""N = 1e10 clustered: are the last bin's RECORDS (after k_part) complete?  (debug aid, round 5)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, lime_amd
from lime_amd import _lib
dev = torch.device("cuda", 0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000_000
nr, ng = 1_000_000, 1000
os.environ["LIME_UPDATE_PATH"] = "cas"
c2 = lime_amd.Context()
del os.environ["LIME_UPDATE_PATH"]
lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
c2.synth_dev(42, 0, n, nr, ng, 16, 1, lcp, da, None)
tb = lime_amd.sim_bytes(nr, ng)
T = torch.empty(tb, dtype=torch.uint8, device=dev)
c2.fused_dev(lcp, da, None, n, n, True, nr, ng, 16, T, True)
s, rc = c2.stats(); print("cas rc", rc, s.n_updates, flush=True)
c = lime_amd.Context()
c.fused_records_dev(lcp, da, None, n, n, True, nr, ng, 16)
s, rc = c.stats(); print("records rc", rc, s.n_updates, "wrm", s.wave_records_max, flush=True)
r, base = c.records_get()
nb, sh = r.n_bins, r.bin_shift
print("bins", nb, "shift", sh, "total", int(base[nb]), "bigrecs", r.n_bigrecs)
bad = 0
for b in list(range(0, nb, 53)) + [nb - 3, nb - 2, nb - 1]:
    lo, hi = int(base[b]), int(base[b + 1])
    recs = torch.empty(hi - lo, dtype=torch.int32, device=dev)
    assert _lib.hip_memcpy_d2d(recs.data_ptr(), r.d_recs + 4 * lo, 4 * (hi - lo)) == 0
    rr = recs.to(torch.int64) & 0xFFFFFFFF
    off = rr & ((1 << sh) - 1); t = rr >> sh
    cells = min(1 << sh, tb - (b << sh))
    h = torch.bincount(off[t > 0], weights=t[t > 0].double(), minlength=1 << sh)[:cells]
    exp = T[b << sh:(b << sh) + cells].double()
    d = int((h.remainder(256) != exp).sum())
    zero = int((t == 0).sum())
    print(f"bin {b}: {hi - lo} records, {zero} empty slots, cells differing from the cas table: {d}", flush=True)
    if d:
        ix = (h.remainder(256) != exp).nonzero().flatten()[:10]
        print("   cells", ix.tolist(), "records say", h[ix].tolist(), "table", exp[ix].tolist())
        # where in the bin's range are the empty slots?
        z = (t == 0).nonzero().flatten()
        if z.numel(): print("   empty slots at", z[:5].tolist(), "..", z[-5:].tolist(), "count", z.numel())
