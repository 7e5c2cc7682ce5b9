#!/bin/bash
# tools/r04_pmc_lines.sh -- HBM bytes and LDS conflict cycles of the after-scan kernels with 477 first-level bins (1 GB table), 2e9 symbols
export TMPDIR=/tmp
for set in "WRITE_SIZE" "FETCH_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"; do
  echo "set: $set"; OUT=/tmp/pmcl; rm -rf $OUT
  C3_PATHS=bin C3_N=2000000000 C3_NR=1000000 C3_NG=1000 rocprofv3 --pmc $set --output-format csv -d $OUT -- python3 tools/bench_c3.py > /tmp/pmcl.log 2>&1
  f=$(find $OUT -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
per = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])): per[(r["Dispatch_Id"], r["Kernel_Name"].split("(")[0], r["Counter_Name"])] += float(r["Counter_Value"])
for (d, k, c), v in per.items(): acc[k][c].append(v)
for k, cs in acc.items():
    if any(x in k for x in ("k_part", "k_sort_tiles", "k_apply_tiles")):
        print(k[-24:], " ".join("%s %.5g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
PY
done
