#!/bin/bash
# tools/variants.sh -- bench each prebuilt variants/lib_*.so in place of lime_amd/liblime_hip.so (timing only)
cp lime_amd/liblime_hip.so /tmp/lib_keep.so
for v in variants/lib_*.so; do
  cp $v lime_amd/liblime_hip.so
  for m in 0 1; do
  echo -n "$v mode $m  "
  python3 bench.py --steps 20 --warmup 3 --no-cpu --mode $m 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('k_scan ms', round(d['roofline']['kernel_ms_avg'],4), 'step ms', round(d['ms_per_step'],4), 'upd', d['config']['table_updates'])"
  done
done
cp /tmp/lib_keep.so lime_amd/liblime_hip.so
