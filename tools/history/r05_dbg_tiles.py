#!/usr/bin/env python3
"""small reproduction: one bin of 32 (or 27 of 32) regions with 100 .. 700 second-level tiles, against the oracle (debug aid, round 5)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, lime_amd
from oracle import oracle_py as O
os.environ["LIME_UPDATE_PATH"] = "bin"; os.environ["LIME_BIN_LEVELS"] = "1,1"
for nr, ng in ((2048, 1024), (2048, 864)):
  for wide in ("0", "1"):
    os.environ["LIME_APPLY_WIDE"] = wide
    c = lime_amd.Context()
    for n in (9_000_000, 12_000_000, 14_000_000, 16_000_000, 17_700_000, 20_000_000, 22_000_000):
        lcp, da, _ = O.synth(77, 0, n, nr, ng, 16, 1)
        cl, nc, ml = O.detect(lcp, da, nr, 16)
        exp = O.score(da, None, cl, nr, ng, threads=8)
        bad = []
        for rep in range(3):
            sim, gnc, gml = c.fused(lcp, da, None, nr, ng, 16)
            s, rc = c.stats()
            bad.append(int((sim != exp).sum()))
        print(f"{nr}x{ng} wide={wide} n={n}: updates {s.n_updates} (~{s.n_updates // 8192} tiles) rc={rc} differing cells {bad}", flush=True)
    c.close()
