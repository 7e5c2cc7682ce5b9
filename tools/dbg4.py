import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lime_amd
from oracle import oracle_py as O
ctx = lime_amd.Context()
for n in [int(x) for x in sys.argv[1:]]:
    lcp, da, eb = O.synth(3, 0, n, 50, 7, 16, 1)
    cl, nc, ml = O.detect(lcp, da, 50, 16)
    gcl, gnc, gml = ctx.detect(lcp, da, 50, 16)
    m = min(len(gcl), len(cl))
    bad = np.nonzero((gcl[:m] != cl[:m]).any(axis=1))[0]
    print(n, "oracle", nc, ml, "detect", gnc, gml, "first bad idx", bad[:3].tolist(), "gpu", gcl[bad[:3]].tolist(), "exp", cl[bad[:3]].tolist())
