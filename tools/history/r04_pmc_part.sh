#!/bin/bash
# tools/r04_pmc_part.sh -- what the partition kernel waits for: address translation, the store path, the memory side (separate --pmc passes)
export TMPDIR=/tmp
N=${1:-10000000000}; NR=${2:-1000000}; NG=${3:-1000}
for set in "TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_PENDING_STALL_CYCLES" "TA_TA_BUSY TCP_TCC_WRITE_REQ TCP_TCC_READ_REQ TCP_TOTAL_WRITE" "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TOO_MANY_EA_WRREQS_STALL"; do
  echo "set: $set"; OUT=/tmp/pmcq; rm -rf $OUT
  C3_PATHS=bin C3_N=$N C3_NR=$NR C3_NG=$NG rocprofv3 --pmc $set --output-format csv -d $OUT -- python3 tools/bench_c3.py > /tmp/pmcq.log 2>&1
  f=$(find $OUT -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("no counters:", e); sys.exit(0)
per = collections.defaultdict(float)
for r in rows: per[(r["Dispatch_Id"], r["Kernel_Name"].split("(")[0], r["Counter_Name"])] += float(r["Counter_Value"])
for (d, k, c), v in per.items(): acc[k][c].append(v)
for k, cs in acc.items():
    if any(x in k for x in ("k_part", "k_sort_tiles", "k_apply_tiles")):
        print(k[-24:], " ".join("%s %.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
PY
done
