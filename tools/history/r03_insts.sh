#!/bin/bash
# tools/r03_insts.sh lib... -- total instructions (all categories) and busy cycles of k_scan on configs[2] (binned), per library build
export TMPDIR=/tmp
cp lime_amd/liblime_hip.so /tmp/lib_keep.so
for lib in "$@"; do
  cp $lib lime_amd/liblime_hip.so
  echo "== $lib"
  C3_PATHS=bin bash tools/pmc_c3.sh "SQ_INSTS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" 'k_scan<'
  grep '^{' gpurun_out/pmc_c3/log.txt | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('scan ms under pmc', d['bin']['parts_ms']['scan'])"
done
cp /tmp/lib_keep.so lime_amd/liblime_hip.so
