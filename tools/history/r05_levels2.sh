#!/bin/bash
# configs[2]'s table (5 GB) in 299 bins of 256 regions (k_part_lines fits) against its 1193 of 64 (k_part)
for lv in "" "512,512" "600,600"; do
  echo "== LIME_BIN_LEVELS=$lv"
  LIME_BIN_LEVELS=$lv python3 tools/r05_probe.py 1e9,1000000,5000,0,0 1e9,1000000,5000,0,1 2>&1 | grep -v amdgpu.ids
done
