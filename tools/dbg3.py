import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lime_amd
from oracle import oracle_py as O
ctx = lime_amd.Context()
rng = np.random.default_rng(1)
for n in (192, 250, 256, 258, 300, 320, 384, 448, 576, 640):
    nr = ng = 30000
    lcp = np.full(n, 20, np.uint32); lcp[::2] = 0
    # unique docs so that each cluster maps to a unique cell
    da = np.empty(n, np.uint32)
    isr = rng.random(n) < 0.5
    da[isr] = np.arange(isr.sum()); da[~isr] = nr + np.arange((~isr).sum())
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    exp = O.score(da, None, cl, nr, ng)
    sim, gnc, gml = ctx.fused(lcp, da, None, nr, ng, 16)
    st, _ = ctx.stats()
    print('   n_med', list(st.n_med), 'n_upd', st.n_updates, 'exp upd', int((exp>0).sum()))
    miss = []
    for ps, ln in cl:
        seg = da[ps:ps+ln]; r = seg[seg < nr][0]; g = seg[seg >= nr][0] - nr
        if sim[r, g] != exp[r, g]: miss.append(int(ps))
    extra = np.argwhere((sim != exp) & (exp == 0))
    print(n, "clusters", nc, "missing heads", miss, "mod64", [m % 64 for m in miss], "extra cells", extra.tolist()[:6])
    for r, g in extra[:6]:
        pr = int(np.nonzero(da == r)[0][0]); pg = int(np.nonzero(da == g + nr)[0][0])
        print("   extra pair read pos", pr, "genome pos", pg)
