#!/usr/bin/env python3
"""tools/dbg_repeat.py -- the same fused pass several times: counters and table must repeat (debug aid for races in the scan)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, lime_amd
n, nr, ng, alpha = int(os.environ.get("N", 333_333_333)), 1_000_000, 5000, 16
dev = torch.device("cuda:0")
lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
os.environ["LIME_UPDATE_PATH"] = os.environ.get("PATHK", "cas")
ctx = lime_amd.Context()
ctx.synth_dev(42, 0, n, nr, ng, alpha, 0, lcp, da, None)
tb = lime_amd.sim_bytes(nr, ng)
ref = None
for it in range(int(os.environ.get("REPS", 8))):
    A = torch.empty(tb, dtype=torch.uint8, device=dev)
    ctx.fused_dev(lcp, da, None, n, n, True, nr, ng, alpha, A, True)
    s, rc = ctx.stats(); assert rc == 0
    if ref is None: ref = A; eq = True
    else: eq = bool(torch.equal(ref, A)); 
    print(it, s.n_clusters, s.max_len, s.n_updates, "table equal" if eq else "TABLE DIFFERS: %d cells" % int((ref != A).sum()))
    if os.environ.get("MARK"):
        import ctypes, numpy as np
        buf = np.zeros(1 << 20, dtype=np.uint32)
        lime_amd._lib.load().lime_debug_winmark(buf.ctypes.data_as(ctypes.c_void_p))
        nt = (n + 1023) // 1024
        m = buf[:nt]; cnt = m >> 16
        lost = np.nonzero(cnt == 0)[0]; dup = np.nonzero(cnt > 1)[0]
        print("   windows %d lost %d dup %d; lost:" % (nt, len(lost), len(dup)), lost[:40].tolist(), "dup:", dup[:20].tolist())
        for w in lost[:3]:
            print("   around lost window %d (chunk %d): takers" % (w, w // 16), [(int(x & 0xFFFF) - 1) // 16 for x in buf[max(0, w - 20):w + 36]])
    if ref is not A: del A
