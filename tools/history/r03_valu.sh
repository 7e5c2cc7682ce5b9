#!/bin/bash
# tools/r03_valu.sh lib... -- vector / scalar / LDS instructions per 1024-position window of the binned EBWT=0 scan (configs[2]),
# one rocprofv3 --pmc pass per library: a deterministic measure where the kernel is bound by instruction issue
export TMPDIR=/tmp
cp lime_amd/liblime_hip.so /tmp/lib_keep.so
for lib in "$@"; do
  cp $lib lime_amd/liblime_hip.so
  OUT=/tmp/valu; rm -rf $OUT
  C3_PATHS=bin C3_N=${C3_N:-1000000000} C3_NG=${C3_NG:-5000} C3_EBWT=${C3_EBWT:-0} rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM --output-format csv -d $OUT -- python3 tools/bench_c3.py > /dev/null 2>&1
  f=$(find $OUT -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$(basename $lib)" "${C3_N:-1000000000}" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if "k_scan<" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
w = float(sys.argv[3]) / 1024.0
for k, cs in acc.items():
    n = len(disp[k])
    print("%-14s %s per window: VALU %.1f SALU %.1f LDS %.1f BRANCH %.1f VMEM %.1f" % (sys.argv[2], k.replace("void lime::", ""), cs["SQ_INSTS_VALU"] / n / w, cs["SQ_INSTS_SALU"] / n / w, cs["SQ_INSTS_LDS"] / n / w, cs["SQ_INSTS_BRANCH"] / n / w, cs["SQ_INSTS_VMEM"] / n / w))
PY
done
cp /tmp/lib_keep.so lime_amd/liblime_hip.so
