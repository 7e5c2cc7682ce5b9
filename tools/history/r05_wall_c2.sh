#!/bin/bash
# start/end of every wave of the scan on configs[1] (10^8 symbols, EBWT=1), both update paths (tools/pt_wall.sh; variants/lib_wall.so)
for p in cas bin; do
  echo "== $p"
  if [ $p = cas ]; then wpb=$1; else wpb=$2; fi
  C3_WPB=$wpb C3_PATHS=$p C3_EBWT=1 C3_N=100000000 C3_NR=100000 C3_NG=500 tools/pt_wall.sh
done
