#!/usr/bin/env python3
"""which of {full binned pass, shard sum, compare-and-swap pass} differs at N = 1e10 clustered (debug aid, round 5)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, lime_amd
from lime_amd.dist import shard_ranges
dev = torch.device("cuda", 0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000_000
nr, ng, ebwt, mode = 1_000_000, 1000, 0, 1
c = lime_amd.Context()
lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
c.synth_dev(42, 0, n, nr, ng, 16, mode, lcp, da, None)
tb = lime_amd.sim_bytes(nr, ng)
A = torch.empty(tb, dtype=torch.uint8, device=dev)
c.fused_dev(lcp, da, None, n, n, True, nr, ng, 16, A, True)
sA, rc = c.stats(); print("full: rc", rc, "updates", sA.n_updates, "wrm", sA.wave_records_max, "sum", int(A.sum(dtype=torch.int64)), flush=True)
os.environ["LIME_UPDATE_PATH"] = "cas"
c2 = lime_amd.Context()
del os.environ["LIME_UPDATE_PATH"]
C_ = torch.empty(tb, dtype=torch.uint8, device=dev)
c2.fused_dev(lcp, da, None, n, n, True, nr, ng, 16, C_, True)
sC, rc = c2.stats(); print("cas: rc", rc, "updates", sC.n_updates, "wrm", sC.wave_records_max, "sum", int(C_.sum(dtype=torch.int64)), flush=True)
d = (A != C_)
print("full vs cas: differing cells", int(d.sum()), flush=True)
if int(d.sum()):
    idx = d.nonzero().flatten()
    print(" first", idx[:20].tolist(), "last", idx[-5:].tolist())
    print(" A", A[idx[:20]].tolist(), "C", C_[idx[:20]].tolist())
    bins = (idx >> 21).unique()
    print(" bins (2 MB) touched:", bins.numel(), bins[:40].tolist())
    regs = (idx >> 16).unique()
    print(" regions touched:", regs.numel(), regs[:40].tolist())
S = torch.zeros(tb, dtype=torch.uint8, device=dev); B = torch.empty_like(S)
for lo, hi, hh in shard_ranges(n, 8):
    c.fused_dev(lcp[lo:], da[lo:], None, hi - lo, hh - lo, hh == n, nr, ng, 16, B, True)
    s, rc = c.stats(); assert rc == 0
    S += B
d = (S != C_)
print("shards vs cas: differing cells", int(d.sum()), flush=True)
if int(d.sum()):
    idx = d.nonzero().flatten()
    print(" first", idx[:20].tolist()); print(" S", S[idx[:20]].tolist(), "C", C_[idx[:20]].tolist())
    print(" regions touched:", (idx >> 16).unique().numel())
