#!/bin/bash
# tools/r03_bins.sh -- after-scan time vs number of first-level bins (LIME_BIN_LEVELS=1,N): configs[2] and N = 1e10
export TMPDIR=/tmp
run() { python3 tools/bench_c3.py 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); b=d['bin']['parts_ms']; print('  scan %.3f after %.3f pass %.3f' % (b['scan'], b['after_scan'], b['pass']))"; }
for nb in 2400 1200 600 300 150; do echo "c3 bins<=$nb"; LIME_BIN_LEVELS=1,$nb C3_PATHS=bin run; done
for nb in 1000 480 240 120 60; do echo "n1e10 bins<=$nb"; LIME_BIN_LEVELS=1,$nb C3_N=10000000000 C3_NG=1000 C3_PATHS=bin run; done
