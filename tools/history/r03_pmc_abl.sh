#!/bin/bash
# tools/r03_pmc_abl.sh LIB -- instruction counts by category of the scan cut after each phase (LIME_ABLATE build), configs[2] binned
export TMPDIR=/tmp
cp lime_amd/liblime_hip.so /tmp/lib_keep.so; cp $1 lime_amd/liblime_hip.so
for k in 1 3 4 10 11 0; do
  echo "== ablate=$k"
  LIME_ABLATE=$k C3_PATHS=bin bash tools/pmc_c3.sh "SQ_INSTS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES" 'k_scan<'
done
cp /tmp/lib_keep.so lime_amd/liblime_hip.so
