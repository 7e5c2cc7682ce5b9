#!/bin/bash
# tools/r04_pmc_scan.sh -- the scan kernel's counters per 10^9 symbols at several input sizes (its rate grows with the input: what changes?)
export TMPDIR=/tmp
NR=${NR:-1000000}; NG=${NG:-1000}
for N in "$@"; do
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  echo "N=$N set: $set"; OUT=/tmp/pmcq; rm -rf $OUT
  C3_PATHS=bin C3_N=$N C3_NR=$NR C3_NG=$NG rocprofv3 --pmc $set --output-format csv -d $OUT -- python3 tools/bench_c3.py > /tmp/pmcq.log 2>&1
  f=$(find $OUT -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$N" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("no counters:", e); sys.exit(0)
n = float(sys.argv[2]) / 1e9
per = collections.defaultdict(float)
for r in rows: per[(r["Dispatch_Id"], r["Kernel_Name"].split("(")[0], r["Counter_Name"])] += float(r["Counter_Value"])
for (d, k, c), v in per.items(): acc[k][c].append(v)
for k, cs in acc.items():
    if "k_scan<" in k:
        print(k[-24:], "per 1e9 symbols:", " ".join("%s %.4g" % (c, sum(v) / len(v) / n) for c, v in sorted(cs.items())))
PY
done
done
