#!/bin/bash
# tools/ab_bench.sh lib... -- bench.py per workload (WLS, default "c3 c2 text_tiled") for several builds of the library inside ONE run
cp lime_amd/liblime_hip.so /tmp/lib_keep.so
for rep in 1 2; do
for lib in "$@"; do
  cp $lib lime_amd/liblime_hip.so
  for wl in ${WLS:-c3 c2 text_tiled}; do
    echo -n "$lib $wl (rep $rep): "
    python3 bench.py --workload $wl --no-also --no-cpu --steps 10 --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step ms', round(d['ms_per_step'],4), 'scan ms', round(d['roofline']['kernel_ms_avg'],4), 'frac', round(d['roofline']['frac'],3))"
  done
done
done
cp /tmp/lib_keep.so lime_amd/liblime_hip.so
