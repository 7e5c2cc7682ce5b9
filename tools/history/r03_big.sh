#!/bin/bash
# tools/r03_big.sh -- scan / pass times at the shapes of BASELINE.json configs[3] and configs[4] (EBWT=1, tables beyond 8 GB: 5 and 3 sub-regions)
export TMPDIR=/tmp
run() { python3 tools/bench_c3.py 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); b=d['bin']; p=b['parts_ms']; n=d['symbols']
print('  scan %.3f ms = %.0f GB/s (%.3f of 8 TB/s)  after %.3f  pass %.3f  updates %d' % (p['scan'], 9*n/p['scan']/1e6, 9*n/p['scan']/1e6/8000, p['after_scan'], p['pass'], d['table_updates_bin']))"; }
echo "configs[3] shape: 2e9 symbols, 20249373 x 930 (18.8 GB table)"; C3_EBWT=1 C3_N=2000000000 C3_NR=20249373 C3_NG=930 C3_PATHS=bin run
echo "configs[4] shape: 1e10 symbols, 3000000 x 3423 (10.3 GB table)"; C3_EBWT=1 C3_N=10000000000 C3_NR=3000000 C3_NG=3423 C3_PATHS=bin run
echo "EBWT=1, 1e10 symbols, 1000000 x 1000 (1 GB table)"; C3_EBWT=1 C3_N=10000000000 C3_NR=1000000 C3_NG=1000 C3_PATHS=bin run
if [ "${MORE:-0}" = "1" ]; then
runc() { python3 tools/bench_c3.py 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=[x for x in d if x in ('bin','cas')][0]; p=d[k]['parts_ms']; n=d['symbols']
print('  [%s] scan %.3f ms (%.3f of 8 TB/s)  after %.3f  pass %.3f' % (k, p['scan'], 9*n/p['scan']/1e6/8000, p['after_scan'], p['pass']))"; }
echo "configs[4] shape, binned (pool sized for 0.05 records per symbol)"; LIME_POOL_DENSITY=0.05 C3_EBWT=1 C3_N=10000000000 C3_NR=3000000 C3_NG=3423 C3_PATHS=bin runc
echo "EBWT=1, 1e10 symbols, 1 GB table, compare-and-swap"; C3_EBWT=1 C3_N=10000000000 C3_NR=1000000 C3_NG=1000 C3_PATHS=cas runc
echo "configs[3] shape, compare-and-swap"; C3_EBWT=1 C3_N=2000000000 C3_NR=20249373 C3_NG=930 C3_PATHS=cas runc
fi
