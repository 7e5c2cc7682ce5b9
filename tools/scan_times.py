#!/usr/bin/env python3
"""tools/scan_times.py WORKLOAD[,WORKLOAD..] [rounds] [key=value ..] -- scan and pass time of bench.py's workloads as the library stands (binned path, no probe):
the quick look after a kernel change, before tools/ab_option.py or a full bench run."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import lime_amd  # noqa: E402
from lime_amd import dist as ldist  # noqa: E402

names = sys.argv[1].split(",")
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
opts = dict(kv.split("=", 1) for kv in sys.argv[3:])
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for name in names:
    wl = bench.WORKLOADS[name]
    scan, pas = [], []
    for _ in range(rounds):
        r = bench.run_pass_series(torch, lime_amd, ldist, wl, wl["n"], 5 if wl["n"] >= 10_000_000_000 else 20, 2, 1, 0, dev, None, overlap=False, options=opts or None)
        scan.append(round(r["parts"]["scan"], 4)); pas.append(round(r["pass_ms_each"]["median"], 4))
        parts = {k: round(v, 3) for k, v in r["parts"].items()}
        del r
    print(name, json.dumps({"scan_ms": scan, "pass_ms": pas, "parts": parts}), flush=True)
