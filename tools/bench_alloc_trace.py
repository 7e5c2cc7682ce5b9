#!/usr/bin/env python3
"""tools/bench_alloc_trace.py [bench.py arguments] -- bench.py with the library's option debug_alloc on: every device allocation of the library goes to stderr
(size, time in release / acquire, free memory before and after), to see which allocation of a default run met pages that the driver had to clear."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lime_amd  # noqa: E402
import bench  # noqa: E402

c = lime_amd.Context()
c.set_option("debug_alloc", "1")      # (process-wide)
c.close()
sys.argv = ["bench.py"] + sys.argv[1:]
bench.main()
