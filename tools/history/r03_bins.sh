#!/bin/bash
# tools/r03_bins.sh -- per-kernel times (rocprofv3 averages, us) vs number of first-level bins (LIME_BIN_LEVELS=1,N): configs[2] and N = 1e10
export TMPDIR=/tmp
run() {
  OUT=/tmp/trb; rm -rf $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/bench_c3.py > /dev/null 2>&1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  python3 -c "
import csv
r={x['Name'].split('(')[0].replace('void ','').replace('lime::',''):float(x['AverageNs'])/1e3 for x in csv.DictReader(open('$f'))}
k={a:b for a,b in r.items() if a.startswith('k_') and 'synth' not in a and b > 8}
print('  ', ' '.join('%s %.0f' % (a, b) for a, b in k.items()), ' after-scan sum %.0f' % sum(b for a, b in k.items() if 'k_scan<' not in a))"
}
for nb in ${C3BINS:-1200 600 300 150}; do echo "c3 bins<=$nb"; LIME_BIN_LEVELS=1,$nb C3_PATHS=bin run; done
for nb in ${N10BINS:-480 240 120 60}; do echo "n1e10 bins<=$nb"; LIME_BIN_LEVELS=1,$nb C3_N=10000000000 C3_NG=1000 C3_PATHS=bin run; done
