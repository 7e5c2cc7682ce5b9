#!/bin/bash
# tools/r04_bins.sh -- round-3 kernels (k_part + k_sort_tiles + k_apply_tiles) with fewer first-level bins and several partition
# workgroups per CU: per-kernel times at the shapes of configs[2] and N = 1e10
export TMPDIR=/tmp LIME_SECOND_LEVEL=tiles
for shape in "1000000000 1000000 5000" "10000000000 1000000 1000"; do
  set -- $shape
  for lv in default 1024,512 1024,256 1024,128; do
    for sp in 1 4; do
      if [ $lv = default ]; then unset LIME_BIN_LEVELS; else export LIME_BIN_LEVELS=$lv; fi
      echo "== N=$1 table ${2}x${3} LIME_BIN_LEVELS=$lv LIME_PART_SPLIT=$sp"
      LIME_PART_SPLIT=$sp C3_N=$1 C3_NR=$2 C3_NG=$3 bash tools/ktrace_c3.sh "k_part|k_sort|k_apply|k_scan<" | grep -v '^{'
    done
  done
done
