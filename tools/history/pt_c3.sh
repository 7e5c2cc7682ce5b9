#!/bin/bash
# tools/pt_c3.sh -- in-kernel phase shares of the scan (variants/lib_pt.so, built with -DLIME_PHASE_TIMING) on configs[2]
cp lime_amd/liblime_hip.so /tmp/lib_keep.so
cp variants/lib_pt.so lime_amd/liblime_hip.so
C3_PATHS=${C3_PATHS:-bin} C3_N=${C3_N:-1000000000} python3 tools/bench_c3.py 2>/dev/null | grep "^blk" | tail -24 | python3 -c "
import sys, re, collections
acc = collections.OrderedDict(); n = 0
for ln in sys.stdin:
    for k, v in re.findall(r'(\w+) (\d+)', ln.split(':', 1)[1]):
        acc[k] = acc.get(k, 0) + int(v)
    n += 1
tot = sum(acc.values())
print('waves', n, {k: f'{100.0 * v / tot:.1f}%' for k, v in acc.items()}, 'cycles/wave', tot // max(n, 1))
"
cp /tmp/lib_keep.so lime_amd/liblime_hip.so
