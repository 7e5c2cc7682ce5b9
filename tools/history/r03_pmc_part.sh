#!/bin/bash
# tools/r03_pmc_part.sh -- HBM bytes of the after-scan kernels at N = 1e10 (1 GB table): FETCH_SIZE, WRITE_SIZE in separate passes
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  OUT=/tmp/pmcp_$c; rm -rf $OUT
  C3_PATHS=bin C3_N=${C3_N:-10000000000} C3_NG=${C3_NG:-1000} rocprofv3 --pmc $c --output-format csv -d $OUT -- python3 tools/bench_c3.py > /dev/null 2>&1
  f=$(find $OUT -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]: acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "k_" in k and "synth" not in k and sum(v) / len(v) > 1000:
        m = sum(v) / len(v); print("%s %-28s avg %.3f GB%s" % (sys.argv[2], k[-28:], m * 1024 * (2 if sys.argv[2] == "FETCH_SIZE" else 1) / 1e9, " (x2: 128-B requests counted as 64)" if sys.argv[2] == "FETCH_SIZE" else ""))
PY
done
