#!/usr/bin/env python3
"""tools/bench_c3.py -- BASELINE.json configs[2] (10^9 symbols, 10^6 reads x 5000 genomes = 5 GB table,
EBWT=0, alpha=16) on one MI355X: one fused pass per update path (LIME_UPDATE_PATH=cas|bin), parts of the pass
from HIP events.  Quick A/B tool; the headline line is bench.py's."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lime_amd

n = int(os.environ.get("C3_N", 1_000_000_000))
nr, ng, alpha = int(os.environ.get("C3_NR", 1_000_000)), int(os.environ.get("C3_NG", 5000)), 16
mode = int(os.environ.get("C3_MODE", 0))
dev = torch.device("cuda:0")
lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
eb = torch.empty(n, dtype=torch.uint8, device=dev) if int(os.environ.get("C3_EBWT", 0)) else None   # C3_EBWT=1 C3_N=100000000 C3_NR=100000 C3_NG=500: configs[1]
bps = 9 if eb is not None else 8
res = {"symbols": n, "table_bytes": nr * ng, "mode": mode}
ref = None
for path in os.environ.get("C3_PATHS", "cas,bin").split(","):
    os.environ["LIME_UPDATE_PATH"] = path
    ctx = lime_amd.Context()
    if ref is None:
        ctx.synth_dev(42, 0, n, nr, ng, alpha, mode, lcp, da, eb)
    sim = torch.empty(lime_amd.sim_bytes(nr, ng), dtype=torch.uint8, device=dev)
    for _ in range(2):
        ctx.fused_dev(lcp, da, eb, n, n, True, nr, ng, alpha, sim, True)
        s, rc = ctx.stats(); assert rc == 0, rc
    ctx.set_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        ctx.fused_dev(lcp, da, eb, n, n, True, nr, ng, alpha, sim, True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    parts, launches = ctx.get_timing_ex(); ctx.set_timing(False)
    s, rc = ctx.stats(); assert rc == 0
    res[path] = {"ms_per_pass": dt * 1e3, "symbols_per_s": n / dt, "parts_ms": parts, "k_scan_GBps": bps * n / parts["scan"] / 1e6,
                 "pass_GBps": bps * n / parts["pass"] / 1e6, "wave_records_max": s.wave_records_max}
    res["n_clusters_" + path], res["table_updates_" + path] = int(s.n_clusters), int(s.n_updates)
    if os.environ.get("C3_WALL"):          # debug library (-DLIME_WALL_TIMING): when every wave of the last scan started and ended
        import ctypes, numpy as np
        from lime_amd import _lib
        buf = np.zeros(2 * 8192, dtype=np.uint64)
        _lib.load().lime_debug_wall(buf.ctypes.data_as(ctypes.c_void_p))
        w = buf.reshape(-1, 2); w = w[w[:, 1] > 0]; st, en = w[:, 0].astype(np.int64), w[:, 1].astype(np.int64)
        d = (en - st.min()) / 100.0
        print("waves %d scan %.1f us; start spread %.1f us; end min %.1f mean %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f us" % (
            len(w), parts["scan"] * 1e3, (st.max() - st.min()) / 100.0, d.min(), d.mean(), np.percentile(d, 10), np.median(d), np.percentile(d, 90), d.max()))
        wpb = int(os.environ.get("C3_WPB", 16))
        blk = np.arange(len(w)) // wpb
        for x in range(8):
            dx = d[blk % 8 == x]; print(" xcd %d: end mean %.1f min %.1f max %.1f" % (x, dx.mean(), dx.min(), dx.max()))
        wv = np.arange(len(w)) % wpb
        for x in range(wpb):
            dx = d[wv == x]; print(" wave %d of the workgroup: end mean %.1f min %.1f max %.1f" % (x, dx.mean(), dx.min(), dx.max()))
        nblk = len(w) // wpb
        for x in range(2):
            dx = d[(blk >= nblk // 2) == bool(x)]; print(" workgroups %s: end mean %.1f min %.1f max %.1f" % (["first half", "second half"][x], dx.mean(), dx.min(), dx.max()))
        cu = (blk // 8) % 32
        print(" by CU slot (blk//8 %% 32) mean end:", " ".join("%.0f" % d[cu == x].mean() for x in range(32)))
        bmin = np.array([d[blk == x].min() for x in range(nblk)]); bmax = np.array([d[blk == x].max() for x in range(nblk)])
        order = np.argsort(-bmax)[:12]
        print(" latest workgroups (blk: first wave out .. last wave out):", " ".join("%d:%.0f..%.0f" % (x, bmin[x], bmax[x]) for x in order))
        print(" spread inside a workgroup: mean %.1f max %.1f us; histogram of wave ends (50 us bins from min):" % ((bmax - bmin).mean(), (bmax - bmin).max()),
              np.histogram(d, bins=np.arange(d.min(), d.max() + 50, 50))[0].tolist())
        print(" first 32 waves end:", " ".join("%.0f" % x for x in d[:32]))
    if ref is None:
        ref = sim.clone()
    else:
        res["tables_equal"] = bool(torch.equal(ref, sim))
    ctx.close(); del sim
print(json.dumps(res))
