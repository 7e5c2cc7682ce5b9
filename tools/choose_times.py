#!/usr/bin/env python3
"""tools/choose_times.py [WORKLOAD ..] -- bench.py's choose_flow (scan + clusterAnalyze + clusterChoose with and without the table) alone."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import lime_amd  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for name in (sys.argv[1:] or ["c3", "c4_shape"]):
    r = bench.choose_flow(torch, lime_amd, dev, name)
    print(name, json.dumps({k: ({a: b for a, b in v.items() if a in ("with_table", "without_table", "speedup_median", "pairs")} if isinstance(v, dict) else v)
                            for k, v in r.items() if k.startswith("beta")}), flush=True)
