#!/bin/bash
# N = 1e10's 1 GB table: 477 bins of 32 regions (the rule) against 239 of 64 and 120 of 128 at the first level
for lv in "" "256,256" "128,128"; do
  echo "== LIME_BIN_LEVELS=$lv"
  LIME_BIN_LEVELS=$lv python3 tools/r05_probe.py 1e10,1000000,1000,0,0 1e10,1000000,1000,0,1 2>&1 | grep -v amdgpu.ids
done
