#!/bin/bash
# tools/r03_check.sh [tag] -- GPU parity suite, then configs[2] (binned) and configs[1] (cas) pass timings of the current library
TAG=${1:-chk}
O=gpurun_out/r03_$TAG; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
C3_PATHS=bin python3 tools/bench_c3.py > $O/c3.json 2>$O/c3.err && python3 -c "
import json; d=json.load(open('$O/c3.json')); b=d['bin']; print('c3 scan ms %.3f pass %.3f after %.3f GB/s %.0f frac %.3f' % (b['parts_ms']['scan'], b['parts_ms']['pass'], b['parts_ms']['after_scan'], b['k_scan_GBps'], b['k_scan_GBps']/8000))"
C3_EBWT=1 C3_N=100000000 C3_NR=100000 C3_NG=500 C3_PATHS=cas python3 tools/bench_c3.py > $O/c2.json 2>$O/c2.err && python3 -c "
import json; d=json.load(open('$O/c2.json')); b=d['cas']; print('c2 scan ms %.4f pass %.4f GB/s %.0f frac %.3f' % (b['parts_ms']['scan'], b['parts_ms']['pass'], b['k_scan_GBps'], b['k_scan_GBps']/8000))"
