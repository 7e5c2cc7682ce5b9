#!/usr/bin/env python3
"""bisect the N = 1e10 clustered mismatch: sizes and kernel variants (debug aid, round 5)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, lime_amd
dev = torch.device("cuda", 0)
nr, ng = 1_000_000, 1000
tb = lime_amd.sim_bytes(nr, ng)
def run(n, env):
    for k, v in env.items(): os.environ[k] = v
    c = lime_amd.Context()
    for k in env: del os.environ[k]
    return c
nmax = 10_000_000_000
lcp = torch.empty(nmax, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
c0 = run(0, {"LIME_UPDATE_PATH": "cas"})
c0.synth_dev(42, 0, nmax, nr, ng, 16, 1, lcp, da, None)
A = torch.empty(tb, dtype=torch.uint8, device=dev); T = torch.empty_like(A)
for n, env in [(9_000_000_000, {}), (9_600_000_000, {}), (10_000_000_000, {"LIME_APPLY_WIDE": "0"}), (10_000_000_000, {"LIME_SORT_NT": "0"}),
               (10_000_000_000, {"LIME_PART_LINES": "0"}), (10_000_000_000, {"LIME_FORCE_P64": "1"}), (10_000_000_000, {"LIME_SECOND_LEVEL": "sweeps"})]:
    c0.fused_dev(lcp, da, None, n, n, True, nr, ng, 16, T, True)
    s0, rc = c0.stats(); assert rc == 0
    c = run(n, env)
    c.fused_dev(lcp, da, None, n, n, True, nr, ng, 16, A, True)
    s, rc = c.stats()
    d = (A != T)
    nd = int(d.sum())
    msg = ""
    if nd:
        idx = d.nonzero().flatten()
        msg = f" regions {(idx >> 16).unique().tolist()[:6]}.. lost {int(T.sum(dtype=torch.int64)) - int(A.sum(dtype=torch.int64))}"
    print(f"n={n} env={env}: rc={rc} updates={s.n_updates} wrm={s.wave_records_max} differing cells {nd}{msg}", flush=True)
    c.close()
