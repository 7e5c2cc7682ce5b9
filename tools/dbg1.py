import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lime_amd
from oracle import oracle_py as O
n, nr, ng = 400000, 300, 20
lcp, da, eb = O.synth(5, 0, n, nr, ng, 16, 0)
cl, nc, ml = O.detect(lcp, da, nr, 16)
exp = O.score(da, eb, cl, nr, ng, threads=4)
ctx = lime_amd.Context()
try:
    gcl, gnc, gml = ctx.detect(lcp, da, nr, 16)
    print("detect", gnc, nc, gml, ml, np.array_equal(gcl, cl))
    if not np.array_equal(gcl, cl):
        m = min(len(gcl), len(cl))
        bad = np.nonzero((gcl[:m] != cl[:m]).any(axis=1))[0]
        print("first diffs", bad[:5], gcl[bad[:5]], cl[bad[:5]])
except Exception as e:
    print("detect err", e)
try:
    exp0 = O.score(da, None, cl, nr, ng, threads=4)
    sim, gnc, gml = ctx.fused(lcp, da, None, nr, ng, 16)
    print("fused e0", gnc, nc, gml, ml, np.array_equal(sim, exp0))
except Exception as e:
    print("fused e0 err", e)
    s, rc = ctx.stats()
    print(s.n_clusters, s.max_len, s.n_cross, s.n_big, s.flags)
try:
    sim, gnc, gml = ctx.fused(lcp, da, eb, nr, ng, 16)
    print("fused", gnc, nc, gml, ml, np.array_equal(sim, exp), int((sim != exp).sum()))
except Exception as e:
    print("fused err", e)
    s, rc = ctx.stats()
    print(s.n_clusters, s.max_len, s.n_cross, s.n_big, s.flags)
