#!/usr/bin/env python3
"""tools/dbg_golden.py GOLDEN [GOLDEN...] -- the fused pass on golden vectors, both builds and both update paths: cluster count and table
against the reference's outputs (debug aid; run once per library build: tools/dbg_libs.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
for name in sys.argv[1:]:
    g = np.load(f"tests/golden/{name}.npz")
    nr, ng, alpha = int(g["params"][0]), int(g["params"][1]), int(g["params"][2])
    for path in ("bin", "cas"):
        os.environ["LIME_UPDATE_PATH"] = path
        import lime_amd
        ctx = lime_amd.Context()
        for e in (1, 0):
            eb = g["ebwt"] if e else None
            try:
                sim, nc, ml = ctx.fused(g["lcp"], g["da"], eb, nr, ng, alpha)
                exp = g[f"sim_e{e}"]
                bad = int((sim != exp).sum())
                print(f"{name} path={path} ebwt={e}: nc {nc} (want {len(g['clrs'])}) max_len {ml} (want {int(g['clrs'][:,1].max()) if len(g['clrs']) else 0}) table cells wrong {bad} sum {int(sim.sum())} want {int(exp.sum())}")
            except Exception as ex:
                print(f"{name} path={path} ebwt={e}: EXC {ex}")
        ctx.close()
# per-window counts (library built with -DLIME_DEBUG_CNT): LIME_DBG_WINDOWS=1
if os.environ.get("LIME_DBG_WINDOWS"):
    import ctypes as C
    from oracle import oracle_py as O
    for name in sys.argv[1:]:
        g = np.load(f"tests/golden/{name}.npz")
        nr, ng, alpha = int(g["params"][0]), int(g["params"][1]), int(g["params"][2])
        os.environ["LIME_UPDATE_PATH"] = "bin"
        ctx = lime_amd.Context()
        n = len(g["lcp"]); nw = (n + 1023) // 1024
        for rep in range(3):
            sim, nc, ml = ctx.fused(g["lcp"], g["da"], g["ebwt"], nr, ng, alpha)
            out = (C.c_uint32 * nw)()
            ctx.lib.lime_debug_tile_counts(ctx.h, out, nw)
            got = np.array(out[:])
            cl = g["clrs"]
            # expected: clusters whose head lies in window w and that close inside window + read-ahead (end <= (w+1)*1024 + 16 ... the next head is at most at WIN+15)
            w = cl[:, 0] // 1024; end = cl[:, 0] + cl[:, 1]
            inside = end <= (w + 1) * 1024 + 15
            inside |= end >= n          # closed by the end of the data
            exp = np.bincount(w[inside].astype(np.int64), minlength=nw)
            diff = np.nonzero(got != exp)[0]
            print(f"{name} rep {rep}: nc {nc} want {len(cl)}; windows differing: {[(int(i), int(got[i]), int(exp[i])) for i in diff[:12]]}")
        ctx.close()
