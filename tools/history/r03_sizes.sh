#!/bin/bash
# tools/r03_sizes.sh -- scan time vs N and table size (binned, EBWT=0): separates the size effect from the sub-region count
export TMPDIR=/tmp
for cfg in "1000000000 1000" "1000000000 5000" "2000000000 5000" "4000000000 5000" "4000000000 1000" "10000000000 5000"; do
  set -- $cfg
  C3_N=$1 C3_NR=1000000 C3_NG=$2 C3_PATHS=bin python3 tools/bench_c3.py 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); b=d['bin']['parts_ms']; n=d['symbols']
print('N=%g NG=$2 scan %.3f ms frac %.3f pass %.3f after %.3f updates %d' % (n, b['scan'], 8*n/b['scan']/1e6/8000, b['pass'], b['after_scan'], d['table_updates_bin']))"
done
