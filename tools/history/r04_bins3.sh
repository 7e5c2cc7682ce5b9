#!/bin/bash
# tools/r04_bins3.sh -- the first level's bin count once more with the final round-4 kernels, N = 1e10 (1 GB table) and configs[4]'s shape
export TMPDIR=/tmp
for shape in "10000000000 1000000 1000 0" "10000000000 3000000 3423 1"; do
  set -- $shape
  for lv in default 1024,256 1024,1024 1024,2048; do
    if [ $lv = default ]; then unset LIME_BIN_LEVELS; else export LIME_BIN_LEVELS=$lv; fi
    echo "== N=$1 table ${2}x${3} EBWT=$4 LIME_BIN_LEVELS=$lv"
    C3_EBWT=$4 C3_N=$1 C3_NR=$2 C3_NG=$3 bash tools/ktrace_c3.sh "k_part|k_sort|k_apply" | grep -v '^{'
  done
done
