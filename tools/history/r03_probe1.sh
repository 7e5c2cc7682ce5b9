#!/bin/bash
# tools/r03_probe1.sh -- round 3, first GPU call: (1) load-pattern microbenchmark, (2) dynamic instruction counts of the scan cut
# after each phase (variants/lib_abl.so, -DLIME_ABLATE_BUILD) for EBWT=0 (configs[2], binned) and EBWT=1 (configs[1], cas).
export TMPDIR=/tmp
O=gpurun_out/r03_probe1; mkdir -p $O
timeout -k 10 200 ./tools/load_bench > $O/load_bench.txt 2>&1; echo "load_bench rc=$?"
cp lime_amd/liblime_hip.so /tmp/lib_keep.so
cp variants/lib_abl.so lime_amd/liblime_hip.so
for k in 1 3 4 10 11 0; do
  export LIME_ABLATE=$k
  echo "== c3 ablate=$k" >> $O/pmc.txt
  C3_PATHS=bin timeout -k 10 300 bash tools/pmc_c3.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" 'k_scan<' >> $O/pmc.txt 2>&1
  grep '^{' gpurun_out/pmc_c3/log.txt | tail -1 >> $O/pmc.txt
  echo "== c2 ablate=$k" >> $O/pmc.txt
  C3_EBWT=1 C3_N=100000000 C3_NR=100000 C3_NG=500 C3_PATHS=cas timeout -k 10 300 bash tools/pmc_c3.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" 'k_scan<' >> $O/pmc.txt 2>&1
  grep '^{' gpurun_out/pmc_c3/log.txt | tail -1 >> $O/pmc.txt
  echo "done ablate=$k"
done
unset LIME_ABLATE
cp /tmp/lib_keep.so lime_amd/liblime_hip.so
C3_PATHS=bin python3 tools/bench_c3.py > $O/c3_plain.json 2>$O/c3_plain.err
C3_EBWT=1 C3_N=100000000 C3_NR=100000 C3_NG=500 C3_PATHS=cas python3 tools/bench_c3.py > $O/c2_plain.json 2>$O/c2_plain.err
tail -3 $O/c3_plain.json $O/c2_plain.json
