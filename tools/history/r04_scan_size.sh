#!/bin/bash
# tools/r04_scan_size.sh -- the scan kernel against the size of the input and the shape of the table: is there a fixed part?
export KREGEX="k_scan<"
for n in 2.5e8 5e8 1e9 2e9 4e9; do
  echo "######## n = $n"
  BENCH_ARGS="--n $n" bash tools/r04_ab.sh "LIME_X=0" c3 n1e10
done
