#!/bin/bash
# tools/r04_bins2.sh -- first-level bin count sweep at N = 1e10 (1 GB table) with the round-4 kernels
export TMPDIR=/tmp
for lv in default 1024,256 1024,1024 1024,2048; do
  if [ $lv = default ]; then unset LIME_BIN_LEVELS; else export LIME_BIN_LEVELS=$lv; fi
  echo "== N=1e10 1000000x1000 LIME_BIN_LEVELS=$lv"
  C3_N=10000000000 C3_NR=1000000 C3_NG=1000 bash tools/ktrace_c3.sh "k_part|k_sort|k_apply|k_scan<" | grep -v '^{'
done
