#!/bin/bash
# tools/r03_ab.sh lib... -- scan / pass times (HIP events) of configs[2] (binned) and configs[1] (cas) for several library builds inside one
# run, two repetitions; then SQ instruction counters of the LAST library.  PMC=0 skips the counters.
export TMPDIR=/tmp
cp lime_amd/liblime_hip.so /tmp/lib_keep.so
for rep in 1 2; do
for lib in "$@"; do
  cp $lib lime_amd/liblime_hip.so
  C3_PATHS=bin python3 tools/bench_c3.py 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); b=d['bin']; print('$lib c3 scan %.3f pass %.3f after %.3f frac %.3f upd %d nc %d' % (b['parts_ms']['scan'], b['parts_ms']['pass'], b['parts_ms']['after_scan'], b['k_scan_GBps']/8000, d['table_updates_bin'], d['n_clusters_bin']))"
  C3_EBWT=1 C3_N=100000000 C3_NR=100000 C3_NG=500 C3_PATHS=cas python3 tools/bench_c3.py 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); b=d['cas']; print('$lib c2 scan %.4f pass %.4f frac %.3f upd %d nc %d' % (b['parts_ms']['scan'], b['parts_ms']['pass'], b['k_scan_GBps']/8000, d['table_updates_cas'], d['n_clusters_cas']))"
done
done
if [ "${PMC:-1}" = "1" ]; then
  CNT="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
  C3_PATHS=bin bash tools/pmc_c3.sh "$CNT" 'k_scan<'
  C3_EBWT=1 C3_N=100000000 C3_NR=100000 C3_NG=500 C3_PATHS=cas bash tools/pmc_c3.sh "$CNT" 'k_scan<'
fi
cp /tmp/lib_keep.so lime_amd/liblime_hip.so
