#!/bin/bash
# tools/ab_c3.sh lib... -- per-kernel times of tools/bench_c3.py (binned path) for several builds of the library inside ONE run
# (boxes and runs differ by several percent); LIME_BIN_LEVELS may be set
cp lime_amd/liblime_hip.so /tmp/lib_keep.so
for rep in 1 2; do
for lib in "$@"; do
  cp $lib lime_amd/liblime_hip.so
  echo "== $lib (rep $rep) LIME_BIN_LEVELS=${LIME_BIN_LEVELS:-default}"
  bash tools/ktrace_c3.sh "${KREGEX:-k_part|k_apply|k_scan<}" | grep -v '^{'
done
done
cp /tmp/lib_keep.so lime_amd/liblime_hip.so
