#!/usr/bin/env python3
"""bisect the N = 1e10 clustered mismatch, part 2: threshold and table shapes (debug aid, round 5)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, lime_amd
dev = torch.device("cuda", 0)
def mk(env):
    for k, v in env.items(): os.environ[k] = v
    c = lime_amd.Context()
    for k in env: del os.environ[k]
    return c
nmax = 10_000_000_000
lcp = torch.empty(nmax, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
c0 = mk({"LIME_UPDATE_PATH": "cas"})
for n, nr, ng, env in [(3_000_000_000, 1_000_000, 1000, {}), (5_000_000_000, 1_000_000, 1000, {}), (7_000_000_000, 1_000_000, 1000, {}), (8_000_000_000, 1_000_000, 1000, {}),
                       (10_000_000_000, 1_048_576, 1000, {}), (10_000_000_000, 1_000_000, 1000, {"LIME_BIN_LEVELS": "1024,1024"}),
                       (10_000_000_000, 1_000_000, 1000, {"LIME_BIN_LEVELS": "1024,300"}), (10_000_000_000, 1_000_000, 999, {})]:
    tb = lime_amd.sim_bytes(nr, ng)
    A = torch.empty(tb, dtype=torch.uint8, device=dev); T = torch.empty_like(A)
    c0.synth_dev(42, 0, n, nr, ng, 16, 1, lcp, da, None)
    c0.fused_dev(lcp, da, None, n, n, True, nr, ng, 16, T, True)
    s0, rc = c0.stats(); assert rc == 0
    c = mk(env)
    c.fused_dev(lcp, da, None, n, n, True, nr, ng, 16, A, True)
    s, rc = c.stats()
    nbins, sh = c.records_layout(nr, ng)
    d = (A != T)
    nd = int(d.sum())
    msg = ""
    if nd:
        idx = d.nonzero().flatten()
        bins = (idx >> sh).unique().tolist()
        msg = f" bins {bins[:8]} of {nbins} (shift {sh}); lost {int(T.sum(dtype=torch.int64)) - int(A.sum(dtype=torch.int64))}"
    print(f"n={n} {nr}x{ng} env={env}: rc={rc} updates={s.n_updates} differing cells {nd}{msg}", flush=True)
    c.close(); del A, T
