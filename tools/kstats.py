#!/usr/bin/env python3
"""tools/kstats.py DIR -- per-kernel average durations (us) from a rocprofv3 --kernel-trace --stats output directory"""
import csv, glob, re, sys
import os
f = max(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)   # (older runs' files may lie beside it)
for row in csv.DictReader(open(f)):
    name = re.sub(r"\(.*", "", row["Name"]).replace("void ", "").replace("lime::", "")
    if len(sys.argv) > 2 and not re.search(sys.argv[2], name):
        continue
    print(f"{name:28s} calls {row['Calls']:>4s}  avg {float(row['AverageNs'])/1e3:10.1f} us  min {float(row['MinNs'])/1e3:10.1f}  max {float(row['MaxNs'])/1e3:10.1f}")
