#!/usr/bin/env python3
"""tools/bench_stream.py [n] -- PCIe-inclusive rate of the host-pointer entry points: the whole
collection starts in HOST memory (pageable numpy / pinned torch), result table ends in host memory.
Prints one JSON line (never the bench.py `value`)."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lime_amd
from lime_amd.api import check

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 400_000_000
nr, ng, alpha = 100000, 500, 16
ctx = lime_amd.Context()
dev = torch.device("cuda:0")
# generate on the device, bring to the host once (pinned)
lcp_d = torch.empty(n, dtype=torch.int32, device=dev); da_d = torch.empty_like(lcp_d)
eb_d = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.synth_dev(42, 0, n, nr, ng, alpha, 0, lcp_d, da_d, eb_d)
torch.cuda.synchronize()
pin = {k: torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for k, t in (("lcp", lcp_d), ("da", da_d), ("eb", eb_d))}
pin["lcp"].copy_(lcp_d); pin["da"].copy_(da_d); pin["eb"].copy_(eb_d)
torch.cuda.synchronize()
del lcp_d, da_d, eb_d
torch.cuda.empty_cache()
sim = np.zeros((nr, ng), dtype=np.uint8)
res = {"symbols": n, "bytes_in": 9 * n}

def run(lp, dp, ep, chunk, stream=True):
    nc, ml = C.c_uint64(0), C.c_uint64(0)
    t0 = time.perf_counter()
    if stream:
        check(ctx.lib.lime_fused_stream(ctx.h, lp, dp, ep, n, nr, ng, alpha, chunk, sim.ctypes.data, C.byref(nc), C.byref(ml)))
    else:
        check(ctx.lib.lime_fused(ctx.h, lp, dp, ep, n, nr, ng, alpha, sim.ctypes.data, C.byref(nc), C.byref(ml)))
    return time.perf_counter() - t0, int(nc.value), int(sim.sum(dtype=np.uint64))

P = (pin["lcp"].data_ptr(), pin["da"].data_ptr(), pin["eb"].data_ptr())
for name, chunk in (("pinned_chunk64Mi", 0), ("pinned_chunk16Mi", 16 << 20), ("pinned_chunk256Mi", 256 << 20)):
    run(*P, chunk)                                   # warm (buffers, scratch)
    t, nc, cs = run(*P, chunk)
    res[name] = {"s": t, "symbols_per_s": n / t, "GBps_in": 9 * n / t / 1e9, "n_clusters": nc, "checksum": cs}
t, nc, cs = run(*P, 0, stream=False)
t, nc, cs = run(*P, 0, stream=False)
res["pinned_unchunked"] = {"s": t, "symbols_per_s": n / t, "GBps_in": 9 * n / t / 1e9, "n_clusters": nc, "checksum": cs}
pg = [pin[k].numpy().copy() for k in ("lcp", "da", "eb")]      # pageable copies
Q = tuple(a.ctypes.data for a in pg)
run(*Q, 0)
t, nc, cs = run(*Q, 0)
res["pageable_chunk64Mi"] = {"s": t, "symbols_per_s": n / t, "GBps_in": 9 * n / t / 1e9, "n_clusters": nc, "checksum": cs}
print(json.dumps(res))
