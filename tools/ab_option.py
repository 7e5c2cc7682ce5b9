#!/usr/bin/env python3
"""tools/r06_inplace_ab.py [workloads] [values] [rounds] [option] -- an option of the scan by value on bench.py's workloads, rounds interleaved.
Round 6: `inplace_min` (the in-place scoring of 2-symbol clusters, dropped: profiles/r06_inplace_ab.txt) and `dense_min` (from how many accepted
clusters a window lists its 2-symbol clusters apart; 4294967295 = never)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import lime_amd  # noqa: E402
from lime_amd import dist as ldist  # noqa: E402

names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["text_spread", "text_tiled", "c2_clustered", "c2", "c3"]
mins = sys.argv[2].split(",") if len(sys.argv) > 2 else ["4294967295", "64", "128", "192"]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
opt = sys.argv[4] if len(sys.argv) > 4 else "dense_min"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
res = {}
for name in names:
    wl = bench.WORKLOADS[name]
    for rd in range(rounds):
        for m in mins:
            k = 5 if wl["n"] >= 10_000_000_000 else 20
            r = bench.run_pass_series(torch, lime_amd, ldist, wl, wl["n"], k, 2, 1, 0, dev, None, overlap=False, options={opt: m})
            e = res.setdefault(name, {}).setdefault(m, {"scan_ms": [], "pass_ms": [], "updates": r["updates"], "clusters": r["n_clusters"]})
            e["scan_ms"].append(round(r["parts"]["scan"], 4)); e["pass_ms"].append(round(r["pass_ms_each"]["median"], 4))
            assert e["updates"] == r["updates"] and e["clusters"] == r["n_clusters"], (name, m, e, r["updates"], r["n_clusters"])
            del r
    print(name, json.dumps(res[name]), flush=True)
