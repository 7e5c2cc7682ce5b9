#!/bin/bash
# bins of the first partition level: default (<= 2048) against <= 3072 and <= 1024 (LIME_BIN_LEVELS), steady pass parts
for lv in "" "1024,3072" "1024,1024"; do
  echo "== LIME_BIN_LEVELS=$lv"
  LIME_BIN_LEVELS=$lv python3 tools/r05_probe.py 1e10,3000000,3423,1,1 1e10,3000000,3423,1,0 2e9,20249373,930,1,0 1e9,1000000,5000,0,0 2>&1 | grep -v amdgpu.ids
done
