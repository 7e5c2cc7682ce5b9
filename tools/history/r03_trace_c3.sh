#!/bin/bash
# tools/r03_trace_c3.sh lib... -- rocprofv3 kernel averages of tools/bench_c3.py (binned path) per library: configs[2], N = 1e10
export TMPDIR=/tmp
cp lime_amd/liblime_hip.so /tmp/lib_keep.so
for lib in "$@"; do
  cp $lib lime_amd/liblime_hip.so
  for cfg in "1000000000 5000" "10000000000 1000"; do
    set -- $cfg
    OUT=/tmp/tr3; rm -rf $OUT
    C3_PATHS=bin C3_N=$1 C3_NG=$2 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/bench_c3.py > /dev/null 2>&1
    f=$(find $OUT -name "*kernel_stats.csv" | head -1)
    python3 -c "
import csv
r={x['Name'].split('(')[0].replace('void ','').replace('lime::',''):float(x['AverageNs'])/1e3 for x in csv.DictReader(open('$f'))}
print('$(basename $lib) N=$1:', ' '.join('%s %.0f' % (k, v) for k, v in r.items() if k.startswith('k_') and 'synth' not in k and v > 8))"
  done
done
cp /tmp/lib_keep.so lime_amd/liblime_hip.so
