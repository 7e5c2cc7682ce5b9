#!/bin/bash
# tools/r03_part_abl.sh -- k_part with parts of its tile loop cut (variants/lib_abl.so, LIME_ABLATE = 20 no write-out, 21 no
# staging either, 22 no rank atomics either, 23 write-out to consecutive positions): where its time goes.  Results invalid.
export TMPDIR=/tmp
cp lime_amd/liblime_hip.so /tmp/lib_keep.so; cp variants/lib_abl.so lime_amd/liblime_hip.so
for abl in ${ABLS:-0 24}; do
  OUT=/tmp/pabl_$abl; rm -rf $OUT
  LIME_ABLATE=$abl C3_PATHS=bin C3_N=${C3_N:-1000000000} C3_NG=${C3_NG:-5000} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/bench_c3.py > /dev/null 2>&1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  python3 -c "
import csv,sys
r={x['Name'].split('(')[0].replace('void ',''):float(x['AverageNs'])/1e3 for x in csv.DictReader(open('$f'))}
print('ablate $abl:', ' '.join('%s %.1f us' % (k, v) for k, v in r.items() if 'k_part' in k or 'k_scan<' in k))"
done
cp /tmp/lib_keep.so lime_amd/liblime_hip.so
