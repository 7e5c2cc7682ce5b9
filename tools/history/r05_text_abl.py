#!/usr/bin/env python3
"""scan time of the text-derived workload (read ids spread) cut after phase LIME_ABLATE=k (variants/lib_abl.so copied over the library; results
invalid): 1 = loads + staging, 3 = + chunk acceptance, 4 = + cluster list, 10 = + lengths / round bookkeeping, 11 = + 2-4-symbol scoring and the
2-symbol pairs, 0 = everything; 8 = everything but the record drains"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, lime_amd
import bench
dev = torch.device("cuda", 0)
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "text_spread"]
z = np.load(os.path.join(bench.ROOT, "tests", "golden", "text_example.npz"))
k, m = wl["tiled"], len(z["lcp"])
nr1, ng1 = int(z["params"][0]), int(z["params"][1])
n = k * m
l1 = torch.from_numpy(z["lcp"].astype(np.int32)).to(dev); d1 = torch.from_numpy(z["da"].astype(np.int64)).to(dev); e1 = torch.from_numpy(z["ebwt"]).to(dev)
copy = torch.arange(k, device=dev, dtype=torch.int64).repeat_interleave(m)
dd = d1.repeat(k); isr = dd < nr1; rid = dd + copy * nr1
if wl.get("spread"): rid = (rid * int(wl["spread"])) % (k * nr1)
da = torch.where(isr, rid, k * nr1 + copy * ng1 + (dd - nr1)).to(torch.int32)
lcp = l1.repeat(k); eb = e1.repeat(k)
sim = torch.empty(lime_amd.sim_bytes(wl["nr"], wl["ng"]), dtype=torch.uint8, device=dev)
os.environ["LIME_UPDATE_PATH"] = "bin"
c = lime_amd.Context()
for _ in range(3): c.fused_dev(lcp, da, eb, n, n, True, wl["nr"], wl["ng"], 16, sim, True); c.stats()
c.set_timing(True)
for _ in range(10): c.fused_dev(lcp, da, eb, n, n, True, wl["nr"], wl["ng"], 16, sim, True)
p, _ = c.get_timing_ex()
print("ablate=%s scan %.1f us" % (os.environ.get("LIME_ABLATE", "-"), p["scan"] * 1e3), flush=True)
