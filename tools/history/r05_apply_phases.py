#!/usr/bin/env python3
"""where k_apply_tiles' cycles go in mode 0 (table) and mode 1 (row maxima / counts without the table): variants/lib_apt.so (-DLIME_APPLY_TIMING),
configs[2]'s input; per phase, summed over wave 0 of every workgroup.  Run with the instrumented library copied over lime_amd/liblime_hip.so."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, lime_amd
from lime_amd import _lib
dev = torch.device("cuda", 0)
n, nr, ng = 1_000_000_000, 1_000_000, 5000
ctx = lime_amd.Context()
lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
ctx.synth_dev(42, 0, n, nr, ng, 16, 0, lcp, da, None)
names = ["clear LDS + barrier", "first loads issued", "wait for a step's loads", "next step's loads issued", "adds", "barrier after the adds", "index prefetch + write-out / look", "barrier after it"]
out = (ctypes.c_ulonglong * 8)()
lib = _lib.load()
for free in ("0", "1", "0", "1"):
    os.environ["LIME_CHOOSE_FREE"] = free
    lib.lime_debug_part_times(out)                     # (reset)
    ctx.fused_choose_dev(lcp, da, None, n, nr, ng, 16, 85, 0.25)
    lib.lime_debug_part_times(out)
    v = list(out); tot = sum(v) or 1
    print(("without the table (mode 1): " if free == "1" else "with the table (mode 0):    ") + "total %.0f Mcycles; " % (tot / 1e6) +
          "; ".join("%s %.1f%% (%.0f)" % (nm, 100.0 * x / tot, x / 1e6) for nm, x in zip(names, v) if x), flush=True)
