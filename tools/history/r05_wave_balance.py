#!/usr/bin/env python3
"""records of the fullest scan wave against the mean, per bench workload (how even the partition's producers are)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch, lime_amd
from lime_amd import dist as ldist
for name in sys.argv[1:] or ["text_tiled", "c2_clustered", "c3"]:
    wl = bench.WORKLOADS[name]
    r = bench.run_pass_series(torch, lime_amd, ldist, wl, wl["n"], 3, 1, 1, 0, torch.device("cuda", 0), None, overlap=False)
    print(name, "updates", r["updates"], "fullest wave", r["wave_records_max"], "mean per wave", round(r["updates"] / 4096, 1),
          "ratio", round(r["wave_records_max"] / (r["updates"] / 4096), 2), flush=True)
    del r; torch.cuda.empty_cache()
