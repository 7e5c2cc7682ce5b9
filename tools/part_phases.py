#!/usr/bin/env python3
"""tools/part_phases.py WORKLOAD[,..] -- where k_part_lines' cycles go: a library built with -DLIME_PART_TIMING (select it with LIME_LIB) sums, over wave 0 of
every workgroup, the cycles between the kernel's phase marks (PP(i) in lime_kernels.hip):
  hipcc ... -DLIME_PART_TIMING -shared -o variants/lib_ppt.so ...;  LIME_LIB=$PWD/variants/lib_ppt.so python3 tools/part_phases.py n1e10"""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import lime_amd  # noqa: E402
from lime_amd import _lib  # noqa: E402
from lime_amd import dist as ldist  # noqa: E402

NAMES = ["0 scan: read counts, borders", "1 (no barrier)", "2 prefix, descriptors, tasks (2 barriers)", "3 records to their stage slots", "4 barrier",
         "5 count the next tile", "6 lines out + next loads + barrier", "7 tails into the carries"]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for name in sys.argv[1].split(","):
    wl = bench.WORKLOADS[name]
    lib = _lib.load()
    out = (ctypes.c_ulonglong * 8)()
    r = bench.run_pass_series(torch, lime_amd, ldist, wl, wl["n"], 3, 1, 1, 0, dev, None, overlap=False, options={"update_path": "bin", "no_probe": "1"})
    lib.lime_debug_part_times(out)
    v = list(out); tot = sum(v) or 1
    print(name, "after_scan_ms", round(r["parts"]["after_scan"], 3), "; ".join("%s: %.1f%%" % (n, 100.0 * x / tot) for n, x in zip(NAMES, v)), flush=True)
