export TMPDIR=/tmp
OUT=/tmp/icm; rm -rf $OUT
C3_PATHS=bin rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS SQ_WAIT_INST_ANY SQ_IFETCH --output-format csv -d $OUT -- python3 tools/bench_c3.py > /tmp/icm.log 2>&1
f=$(find $OUT -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if "k_scan<" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k, cs in acc.items():
    n = len(disp[k]); print(k, {c: round(v / n / 1e6, 3) for c, v in cs.items()}, "(millions per launch)")
PY
tail -3 /tmp/icm.log | cut -c1-300
